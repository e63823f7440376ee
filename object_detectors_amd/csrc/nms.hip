// YOLOv8 non-max suppression + detection finalisation for gfx950.
//
// Replaces `ops.non_max_suppression` as called at object_detectors/yolov8_lmi/model.py:212 (in-tree
// YOLOv5 analogue object_detectors/yolov5_lmi/trt/yolov5_trt.py:172-282, torchvision nms :267),
// `ops.scale_boxes` + clip (model.py:222; yolov5_trt.py:316-341) and the per-class confidence
// filter (model.py:223-232).  Semantics: oracle/nms.py, oracle/boxes.py.
//
// Kernels
//  nms_decode_filter  grid (anchor tiles, batch): streams the nc class rows of the head once with
//                     128-bit loads (HBM-bound, nc*A*4 bytes/image), class-max / first-argmax per
//                     anchor, strict fp32 `> conf`, wave-ballot compaction of (score, anchor) keys.
//  nms_sort_suppress  one 1024-lane workgroup per image (latency-bound, LDS-resident).  Per-class NMS
//                     with <= 128 classes and <= 2048 candidates: decomposed by class WITHOUT sorting the
//                     list (class_parallel_nms: one lane per candidate, bit rows of suppressing
//                     higher-keyed class members, fixpoint rounds; only the kept keys are sorted).
//                     Otherwise (agnostic, many classes, a class of > 256 candidates, ...): bitonic sort
//                     of the 64-bit keys (score desc, anchor asc = torchvision's stable order), then 64-candidate
//                     chunks, every wave tests the chunk against the kept boxes held in LDS, pair masks
//                     inside the chunk come from wave ballots, the chunk's greedy answer is a fixpoint
//                     in one wave.  Both stop at max_det keeps.
//  finalize_boxes_kernel  one workgroup per image: scale/clip boxes in place, per-class threshold, ranks.
//  finalize_slots_kernel  one workgroup per image: prefix of the keep counts -> dense global mask slots and their inverse map.
#include <hip/hip_fp16.h>

#include <cmath>
#include <mutex>

#include "common.h"

// Workspace of lmi_nms_v8_static (the entry point that allocates nothing): one copy per GPU, held until the process
// ends.  In host code this name is only a handle for hipGetSymbolAddress -- the kernels get the per-device address
// resolved in nms_device_state().
static __device__ __align__(256) unsigned char scratch[LMI_NMS_STATIC_BYTES];

namespace {

constexpr int kSortThreads = 1024;
constexpr int kNmsWindow = 256;   // sorted candidates one step of the greedy sweep looks at (four per lane of a wave)

// Measurement builds only (-DLMI_NMS_STAMPS, scratch/nms_stamps.py): wall-clock stamps of workgroup 0 at the phase
// boundaries of nms_sort_suppress.  The product library is built without it: the macro is empty, nothing is exported.
#ifdef LMI_NMS_STAMPS
__device__ unsigned long long g_nms_stamps[64];
#define NMS_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_nms_stamps[i] = wall_clock64(); } while (0)
extern "C" int lmi_debug_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_nms_stamps), sizeof(g_nms_stamps)); }
extern "C" int lmi_debug_stamps_clear() { static unsigned long long z[64]; return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_nms_stamps), z, sizeof(z)); }
#else
#define NMS_STAMP(i) do {} while (0)
#endif
constexpr int kSortLdsCap = 4096;  // keys sorted inside LDS (32 KiB, so the workgroup can share a CU with other kernels' blocks);
                                   // more candidates than that are sorted in the global workspace

__host__ __device__ inline int next_pow2(int v) {
  int p = 1;
  while (p < v) p <<= 1;
  return p;
}

// float -> uint32 whose unsigned order equals the float order
__device__ __forceinline__ uint32_t f2sortable(float f) {
  uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float sortable2f(uint32_t s) {
  uint32_t u = (s & 0x80000000u) ? (s & 0x7fffffffu) : ~s;
  return __uint_as_float(u);
}

// Head element -> fp32.  fp16 heads (`fp16=True` models, object_detectors/yolov8_lmi/model.py:68,132) are promoted
// value by value -- an exact conversion -- and every comparison and box operation runs in fp32: the same result as
// `pred.float()` followed by the fp32 path, without that extra pass over the head.
__device__ __forceinline__ float ldf(const float *p, size_t i) { return p[i]; }
__device__ __forceinline__ float ldf(const __half *p, size_t i) { return __half2float(p[i]); }
__device__ __forceinline__ void ld4(const float *p, float e[4]) {
  const float4 v = *reinterpret_cast<const float4 *>(p);
  e[0] = v.x; e[1] = v.y; e[2] = v.z; e[3] = v.w;
}
__device__ __forceinline__ void ld4(const __half *p, float e[4]) {
  const uint2 v = *reinterpret_cast<const uint2 *>(p);
  const __half2 lo = *reinterpret_cast<const __half2 *>(&v.x), hi = *reinterpret_cast<const __half2 *>(&v.y);
  e[0] = __low2float(lo); e[1] = __high2float(lo); e[2] = __low2float(hi); e[3] = __high2float(hi);
}

struct NmsLayout {
  size_t off_count, off_keys, off_cls, off_sbox, off_cd, off_flag, total;
  int P;
  int seg;    // anchors per workgroup of the candidate filter = length of its key segment (256, or 64 for odd anchor counts)
  int nblk;   // workgroups (segments) per image
};

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

NmsLayout nms_layout(int batch, int anchors) {
  NmsLayout L;
  L.P = next_pow2(anchors);
  L.seg = (anchors & 3) == 0 ? 256 : 64;
  L.nblk = (anchors + L.seg - 1) / L.seg;
  size_t o = 0;
  // one candidate count per filter workgroup, written unconditionally by every call: nothing has to be cleared
  L.off_count = o; o = align_up(o + sizeof(int) * (size_t)batch * L.nblk, 256);
  L.off_keys = o;  o = align_up(o + sizeof(uint64_t) * (size_t)batch * L.P, 256);
  L.off_cls = o;   o = align_up(o + sizeof(int) * (size_t)batch * anchors, 256);
  L.off_sbox = o;  o = align_up(o + sizeof(float4) * (size_t)batch * anchors, 256);
  L.off_cd = o;    o = align_up(o + sizeof(float2) * (size_t)batch * anchors, 256);  // rotated NMS; with sbox: the suppressor lists
  L.off_flag = o;  o = align_up(o + sizeof(int) * (size_t)batch, 256);               // list form: images left to the sorted sweep
  L.total = o;
  return L;
}

// ---------------------------------------------------------------------------------------------
// Workgroup = 4 waves over the SAME 64*VEC anchors; wave w scans one quarter of the class rows (4x the loads in
// flight of a one-wave-per-anchor-group layout), wave 0 merges the quarters in class order (strict >, so the
// first maximal class wins as in torch.max) and compacts the candidates to the head of the workgroup's OWN key segment
// (64*VEC keys), in anchor order, and writes their number -- zero included -- to the workgroup's own counter: no
// atomics, no counter to clear before the launch, and a candidate order that does not change from run to run.
template <int VEC, typename InT>
__global__ __launch_bounds__(256) void nms_decode_filter(const InT *__restrict__ pred, int nc, int nm, int A, float conf,
                                                        uint64_t *__restrict__ keys, int P, int *__restrict__ cls_of,
                                                        int *__restrict__ blk_count) {
  __shared__ float sbest[3][64 * VEC];
  __shared__ int sidx[3][64 * VEC];
  const int b = blockIdx.y;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int a0 = (blockIdx.x * 64 + lane) * VEC;
  const int C = 4 + nc + nm;
  const InT *cls = pred + ((size_t)b * C + 4) * A;
  const int cpart = (nc + 3) >> 2;
  const int c0 = wave * cpart, c1 = min(nc, c0 + cpart);
  float best[VEC];
  int bi[VEC];
  const bool valid = a0 < A;
#pragma unroll
  for (int k = 0; k < VEC; ++k) { best[k] = -INFINITY; bi[k] = c0; }
  if (valid && c0 < c1) {
    if constexpr (VEC == 4) {
      ld4(cls + (size_t)c0 * A + a0, best);
#pragma unroll 8
      for (int c = c0 + 1; c < c1; ++c) {
        float e[4];
        ld4(cls + (size_t)c * A + a0, e);
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (e[k] > best[k]) { best[k] = e[k]; bi[k] = c; }
      }
    } else {
      best[0] = ldf(cls, (size_t)c0 * A + a0);
#pragma unroll 8
      for (int c = c0 + 1; c < c1; ++c) {
        const float e = ldf(cls, (size_t)c * A + a0);
        if (e > best[0]) { best[0] = e; bi[0] = c; }
      }
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int k = 0; k < VEC; ++k) { sbest[wave - 1][lane * VEC + k] = best[k]; sidx[wave - 1][lane * VEC + k] = bi[k]; }
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w)
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float e = sbest[w][lane * VEC + k];
      if (e > best[k]) { best[k] = e; bi[k] = sidx[w][lane * VEC + k]; }
    }
  int cnt = 0;
  bool flag[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    flag[k] = valid && (a0 + k < A) && (best[k] > conf);
    cnt += flag[k] ? 1 : 0;
  }
  // wave-level exclusive scan of cnt
  int incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    int t = __shfl_up(incl, d);
    if (lane >= d) incl += t;
  }
  const int total = __shfl(incl, 63);
  if (lane == 63) blk_count[(size_t)b * gridDim.x + blockIdx.x] = total;
  if (total == 0) return;
  int o = blockIdx.x * (64 * VEC) + incl - cnt;
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    if (flag[k]) {
      const uint32_t anchor = (uint32_t)(a0 + k);
      keys[(size_t)b * P + o] = ((uint64_t)f2sortable(best[k]) << 32) | (uint64_t)(~anchor);
      cls_of[(size_t)b * A + anchor] = bi[k];
      ++o;
    }
  }
}

// descending bitonic sort of P2 (power of two) keys; all kSortThreads lanes participate
template <typename KeyPtr>
__device__ __forceinline__ void bitonic_desc(KeyPtr a, int P2) {
  const int tid = threadIdx.x;
  for (int k = 2; k <= P2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (P2 >> 1); t += kSortThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int p = i | j;
        const uint64_t x = a[i], y = a[p];
        const bool desc = (i & k) == 0;
        if (desc ? (x < y) : (x > y)) { a[i] = y; a[p] = x; }
      }
      __syncthreads();
    }
  }
}

// The same network for keys in LDS with the strides inside a wave's segment done in REGISTERS: a wave owns a segment
// of 64*KPL keys, lane l holds the keys at segment positions l + 64q.  Strides >= 64 inside the segment pair two
// registers of one lane; strides < 64 pair lanes l and l ^ j (one cross-lane exchange per key, no LDS write / read
// round trip, no fence); only strides that cross segments go through LDS, between two workgroup barriers.  For 2048
// keys: 28 of the 66 steps (stages k <= 128) never touch LDS, 28 more run in registers between the cross-segment
// steps, 10 steps + 14 barriers remain in LDS.
template <int KPL>
__device__ __forceinline__ void bitonic_desc_regs(uint64_t *a, int P2) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int SEG = 64 * KPL;
  const int nseg = P2 / SEG;
  const bool active = wave < nseg;
  const int base = wave * SEG;
  uint64_t r[KPL];
#pragma unroll
  for (int q = 0; q < KPL; ++q) r[q] = active ? a[base + lane + 64 * q] : 0ull;
  for (int k = 2; k <= P2; k <<= 1) {
    int j = k >> 1;
    if (j >= SEG) {   // uniform: strides across segments, every lane of the workgroup on the LDS copy
      if (active) {
#pragma unroll
        for (int q = 0; q < KPL; ++q) a[base + lane + 64 * q] = r[q];
      }
      __syncthreads();
      for (; j >= SEG; j >>= 1) {
        for (int t = tid; t < (P2 >> 1); t += kSortThreads) {
          const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
          const int p = i | j;
          const uint64_t x = a[i], y = a[p];
          const bool desc = (i & k) == 0;
          if (desc ? (x < y) : (x > y)) { a[i] = y; a[p] = x; }
        }
        __syncthreads();
      }
      if (active) {
#pragma unroll
        for (int q = 0; q < KPL; ++q) r[q] = a[base + lane + 64 * q];
      }
    }
    if (active) {
      for (; j >= 64; j >>= 1) {   // both keys of a pair live in this lane
        const int dq = j >> 6;
#pragma unroll
        for (int q = 0; q < KPL; ++q) {
          if ((q & dq) == 0 && (q | dq) < KPL) {
            const bool desc = ((base + lane + 64 * q) & k) == 0;
            const uint64_t x = r[q], y = r[q | dq];
            const uint64_t hi = x > y ? x : y, lo = x > y ? y : x;
            r[q] = desc ? hi : lo;
            r[q | dq] = desc ? lo : hi;
          }
        }
      }
      for (; j > 0; j >>= 1) {     // partner key in lane l ^ j
        const bool low = (lane & j) == 0;
#pragma unroll
        for (int q = 0; q < KPL; ++q) {
          const uint64_t x = r[q];
          const uint64_t y = __shfl_xor((unsigned long long)x, j);
          const bool desc = ((base + lane + 64 * q) & k) == 0;
          const uint64_t hi = x > y ? x : y, lo = x > y ? y : x;
          r[q] = (desc == low) ? hi : lo;
        }
      }
    }
  }
  if (active) {
#pragma unroll
    for (int q = 0; q < KPL; ++q) a[base + lane + 64 * q] = r[q];
  }
  __syncthreads();
}

// descending sort of P2 (power of two) keys in LDS
__device__ __forceinline__ void bitonic_desc_lds(uint64_t *a, int P2) {
  if (P2 >= 128) {
    if (P2 <= 128 * (kSortThreads / 64)) bitonic_desc_regs<2>(a, P2);
    else bitonic_desc_regs<4>(a, P2);   // kSortLdsCap = 4096 keys = 16 waves x 256
    return;
  }
  const int tid = threadIdx.x;           // < 128 keys: a handful of steps, one wave's worth of pairs
  for (int k = 2; k <= P2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      if (tid < (P2 >> 1)) {
        const int i = ((tid & ~(j - 1)) << 1) | (tid & (j - 1));
        const int p = i | j;
        const uint64_t x = a[i], y = a[p];
        const bool desc = (i & k) == 0;
        if (desc ? (x < y) : (x > y)) { a[i] = y; a[p] = x; }
      }
      __syncthreads();
    }
  }
}

__device__ __forceinline__ bool iou_gt(float ax1, float ay1, float ax2, float ay2, float aarea, float bx1, float by1,
                                       float bx2, float by2, float barea, float thr) {
  const float xx1 = fmaxf(ax1, bx1), yy1 = fmaxf(ay1, by1);
  const float xx2 = fminf(ax2, bx2), yy2 = fminf(ay2, by2);
  const float w = fmaxf(0.0f, xx2 - xx1), h = fmaxf(0.0f, yy2 - yy1);
  const float inter = w * h;
  if (!(inter > 0.0f)) return false;  // 0/x and 0/0 never exceed a threshold >= 0
  const float ovr = inter / (aarea + barea - inter);
  return ovr > thr;
}

// ---------------------------------------------------------------------------------------------
// The candidates of image b, compact: the filter left blk_count[blk] keys at the head of segment blk.  All
// kSortThreads lanes call this.  Returns n (0: nothing to do) and sets P2 / in_lds; the keys end up in lkeys[0..P2)
// when they fit the LDS buffer, else in gkeys[0..P2) (through the scratch copy gtmp), padded with zeros.
// scr: 2 * 256 + 32 ints of LDS.
constexpr int kGatherChunk = 256;   // segments whose prefix is held in LDS at a time
constexpr int kRotQueue = 128;      // rotated NMS: near pairs a wave queues before it evaluates 64 of them
constexpr size_t kGatherLds = sizeof(int) * (2 * kGatherChunk + 32);   // gather_candidates' scratch, behind the box buffer

// More candidates than the LDS buffers hold (validation-style thresholds: every anchor a candidate).  The sweep stops
// at max_det keeps, so the whole list is rarely needed: top_keys leaves the (at most) `cap` candidates with the LARGEST
// keys in lkeys, in any order -- the head of the sorted list, exactly, since keys are unique -- found by a radix
// histogram of the score bits (12 bits, then 12 more inside the boundary bin) instead of a sort.  The caller sorts and
// sweeps them in LDS; only if that ends short of max_det keeps does it come back for the whole list (global-memory
// sort, ~400 us).  hist: LDS, see kTopBins.  Returns the number of keys taken (0: the cut fell on a run of more
// than `cap` keys with equal leading 24 score bits -- nothing copied, take the whole list).  All lanes call this.
// number of candidates of an image (what gather_candidates finds first); all lanes call this
__device__ int candidate_total(const int *__restrict__ cnt, int nblk, int A, int *scr) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int part = 0;
  for (int i = tid; i < nblk; i += kSortThreads) part += cnt[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) scr[wave] = part;
  __syncthreads();
  int n = 0;
  for (int w = 0; w < kSortThreads / 64; ++w) n += scr[w];
  __syncthreads();
  return min(n, A);
}

constexpr int kTopBins = 4096;   // hist: kTopBins + 4 + kSortThreads / 64 ints

// smallest bin c with (keys in bins >= c) <= budget; res[0] = c (kTopBins: not even the last bin fits), res[1] = that count
__device__ void suffix_cut(const int *hist, int budget, int *res) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kSortThreads / 64;
  constexpr int PER = kTopBins / kSortThreads;
  int *wsum = res + 4;   // NW ints
  int h[PER], own = 0;
#pragma unroll
  for (int k = 0; k < PER; ++k) { h[k] = hist[PER * tid + k]; own += h[k]; }
  int incl = own;        // inclusive suffix over the lanes of the wave (lanes above me)
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_down(incl, d);
    if (lane + d < 64) incl += u;
  }
  if (lane == 0) wsum[wave] = incl;
  __syncthreads();
  int above = incl - own;
  for (int w = wave + 1; w < NW; ++w) above += wsum[w];
  // bins of this thread from the top down
  int sfx = above;
#pragma unroll
  for (int k = PER - 1; k >= 0; --k) {
    sfx += h[k];                                   // S(c), c = PER * tid + k
    const int c = PER * tid + k;
    const int below = c == 0 ? 0 : (k > 0 ? h[k - 1] : hist[c - 1]);
    if (sfx <= budget && (c == 0 || sfx + below > budget)) { res[0] = c; res[1] = sfx; }
  }
  if (tid == kSortThreads - 1 && h[PER - 1] > budget) { res[0] = kTopBins; res[1] = 0; }
  __syncthreads();
}

__device__ int top_keys(const int *__restrict__ cnt, int nblk, int seg, int cap, uint64_t *lkeys, const uint64_t *gkeys, int *hist) {
  const int tid = threadIdx.x;
  const int shift = __ffs(seg) - 1, slots = nblk << shift;
  int *res = hist + kTopBins;   // [0] cut bin, [1] keys at and above it, [2] copy counter, [4..] suffix_cut's wave sums
  int taken = 0;
  unsigned long long T = 0ull;  // keys whose score word is >= T are taken (2^32: none)
  int bb = 0;                   // boundary bin of level 0
  for (int level = 0; level < 2; ++level) {
    for (int i = tid; i < kTopBins; i += kSortThreads) hist[i] = 0;
    __syncthreads();
    for (int e = tid; e < slots; e += kSortThreads) {
      if ((e & (seg - 1)) < cnt[e >> shift]) {
        const uint32_t hi = (uint32_t)(gkeys[e] >> 32);
        if (level == 0) atomicAdd(&hist[hi >> 20], 1);
        else if ((int)(hi >> 20) == bb) atomicAdd(&hist[(hi >> 8) & 0xfffu], 1);
      }
    }
    __syncthreads();
    suffix_cut(hist, cap - taken, res);
    const int c = res[0], got = res[1];
    __syncthreads();            // res is rewritten by the next level
    if (level == 0) {
      taken = got;
      T = (unsigned long long)c << 20;
      bb = c - 1;               // c >= 1: all the keys together exceed cap
      if (taken >= cap / 2 || bb < 0) break;
    } else if (c < kTopBins) {
      taken += got;
      T = ((unsigned long long)bb << 20) | ((unsigned long long)c << 8);
    }
  }
  if (taken == 0) return 0;
  if (tid == 0) res[2] = 0;
  __syncthreads();
  for (int e = tid; e < slots; e += kSortThreads) {
    if ((e & (seg - 1)) < cnt[e >> shift]) {
      const uint64_t key = gkeys[e];
      if ((unsigned long long)(uint32_t)(key >> 32) >= T) lkeys[atomicAdd(&res[2], 1)] = key;
    }
  }
  __syncthreads();
  return res[2];
}

__device__ int gather_candidates(const int *__restrict__ cnt, int nblk, int seg, int A, int lds_keys, uint64_t *lkeys, uint64_t *gkeys,
                                 uint64_t *gtmp, int *scr, int &P2, bool &in_lds, int *hist, int &n_total, int bail_above = 0x7fffffff) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kSortThreads / 64;
  int *wtot = scr, *ccnt = scr + 32, *cpre = scr + 32 + kGatherChunk;
  int part = 0;
  for (int i = tid; i < nblk; i += kSortThreads) part += cnt[i];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) wtot[wave] = part;
  __syncthreads();
  int n = 0;
  for (int w = 0; w < NW; ++w) n += wtot[w];
  n = min(n, A);
  __syncthreads();
  n_total = n;
  if (n == 0) return 0;
  if (n > bail_above) return -1;   // uniform; nothing has been written
  if (n > lds_keys && hist != nullptr) {   // the head of the list only (see top_keys); hist == nullptr: the caller wants all of it
    const int m = top_keys(cnt, nblk, seg, lds_keys, lkeys, gkeys, hist);
    if (m > 0) {
      P2 = next_pow2(m);
      in_lds = true;
      for (int i = m + tid; i < P2; i += kSortThreads) lkeys[i] = 0ull;
      __syncthreads();
      return m;
    }
  }
  P2 = next_pow2(n);
  in_lds = P2 <= lds_keys;
  uint64_t *dst = in_lds ? lkeys : gtmp;
  const int shift = __ffs(seg) - 1;
  int running = 0;
  for (int base = 0; base < nblk; base += kGatherChunk) {
    const int nb = min(kGatherChunk, nblk - base);
    if (tid < kGatherChunk) {   // waves 0..3: prefix of this chunk's counts
      const int c = tid < nb ? cnt[base + tid] : 0;
      int incl = c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int u = __shfl_up(incl, d);
        if (lane >= d) incl += u;
      }
      if (lane == 63) wtot[wave] = incl;
      ccnt[tid] = c;
      cpre[tid] = incl - c;     // within the wave for now
    }
    __syncthreads();
    int chunk_total = 0;
    for (int w = 0; w < kGatherChunk / 64; ++w) chunk_total += wtot[w];
    for (int e = tid; e < (nb << shift); e += kSortThreads) {
      const int bl = e >> shift, o = e & (seg - 1);
      if (o < ccnt[bl]) {
        int pre = running + cpre[bl];
        for (int w = 0; w < (bl >> 6); ++w) pre += wtot[w];
        if (pre + o < n) dst[pre + o] = gkeys[((size_t)(base + bl) << shift) + o];
      }
    }
    __syncthreads();
    running += chunk_total;
  }
  if (in_lds) {
    for (int i = n + tid; i < P2; i += kSortThreads) lkeys[i] = 0ull;
  } else {
    for (int i = tid; i < P2; i += kSortThreads) gkeys[i] = i < n ? gtmp[i] : 0ull;
  }
  __syncthreads();
  return n;
}

// ---------------------------------------------------------------------------------------------
// Class-parallel suppression, WITHOUT sorting the candidates first.  With per-class NMS the class offset
// (cls * max_wh) moves the boxes of different classes so far apart that they cannot intersect, PROVIDED the x-extent
// of all candidates together is below max_wh (checked per image; the fp32 rounding of the offset coordinates is
// below 0.1 px for <= 128 classes).  torchvision's greedy pass over the score-ordered list then decomposes into
// independent greedy passes per class, each over that class's candidates in the same order and on the same offset
// coordinates -- identical keeps.  Inside a class only the RELATIVE order of two candidates matters, and that is a
// comparison of their keys (score, then anchor: torchvision's stable order).  So the 2048-key bitonic sort of the
// whole list (15 of the 42 us of a dense image) is not needed to decide who is kept:
//   1. the candidates are grouped by class (LDS counting pass; the order inside a class does not matter) and their
//      keys and class-offset boxes are written in grouped order;
//   2. one lane per candidate j tests j against every member of its class with a LARGER key and keeps the outcomes
//      as a bit row (classes of <= 256 candidates: up to four 64-bit words per candidate);
//   3. the greedy answer of every class is the fixpoint of  keep_j = no kept predecessor suppresses j,  reached by
//      rounds over all candidates at once (after t rounds the t best members of every class are final; in practice
//      the length of the longest suppression chain);
//   4. the kept candidates' keys are compacted and only THEY are sorted (a few hundred keys at most) for the
//      score-ordered output and the max_det cut.
// Falls back (returns false; the key list is untouched) when the extent check fails, a class holds more than 256
// candidates, or the largest class is big against the image's candidate count (few large classes: the sorted sweep
// is cheaper, see the bound below); the caller does not come here for agnostic NMS, > kCpMaxClasses classes, > 2048 candidates, more
// candidates than max_nms (that cut needs the global order), or too little LDS.
constexpr int kCpMaxClasses = 128;
constexpr int kCpWords = 4;   // 64-bit words of a class's member set: classes of up to 256 candidates

struct CpScratch {
  int *ccount, *cstart;     // [ncp] candidates per class, first slot of the class in the grouped arrays
  uint64_t *keep_a, *keep_b;  // [ncp][kCpWords] kept members of every class (bit = slot inside the class), two generations
  int *misc;                // [0] largest class, [1..17] wave totals of the compaction, [20] extent verdict
  float *ext;               // [32] per-wave x-extent partials
  unsigned short *lcls;     // [P2] class of candidate i (list order)
  unsigned short *gcls;     // [P2] class of grouped candidate g
  unsigned char *slotof;    // [P2] slot of candidate i inside its class
};

__device__ __forceinline__ size_t cp_bytes(int P2, int ncp) { return (size_t)ncp * (16 * kCpWords + 8) + 32 * 4 + 32 * 4 + (size_t)P2 * 5 + 64; }

__device__ __forceinline__ CpScratch cp_carve(unsigned char *base, int P2, int ncp) {
  CpScratch C;
  unsigned char *q = base;                                    // 16-byte aligned (inside the float4 box buffer)
  C.keep_a = reinterpret_cast<uint64_t *>(q); q += (size_t)ncp * 8 * kCpWords;
  C.keep_b = reinterpret_cast<uint64_t *>(q); q += (size_t)ncp * 8 * kCpWords;
  C.ccount = reinterpret_cast<int *>(q); q += (size_t)ncp * 4;
  C.cstart = reinterpret_cast<int *>(q); q += (size_t)ncp * 4;
  C.misc = reinterpret_cast<int *>(q); q += 32 * 4;
  C.ext = reinterpret_cast<float *>(q); q += 32 * 4;
  C.lcls = reinterpret_cast<unsigned short *>(q); q += (size_t)P2 * 2;
  C.gcls = reinterpret_cast<unsigned short *>(q); q += (size_t)P2 * 2;
  C.slotof = q;
  return C;
}

// All lanes of the workgroup call this (uniform control flow).  keys[0..n): the image's candidate keys in any order
// (n <= 2 * kSortThreads); gkey / gbox: room for n grouped keys / boxes.  On success keys[0..K) holds the kept keys
// in descending order (K = min(kept, max_det)) and *sK = K.
template <typename InT>
__device__ bool class_parallel_nms(const CpScratch &C, uint64_t *keys, uint64_t *gkey, float4 *gbox, const InT *pb, const int *cls_b,
                                   int A, int n, int P2, int ncp, int max_det, float max_wh, float thr, int *sK) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kSortThreads / 64;

  // 1. classes, group by class
  if (tid < ncp) C.ccount[tid] = 0;
  __syncthreads();
  int mycls[2] = {0, 0};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + kSortThreads * u;
    if (i < n) {
      const uint32_t anchor = ~(uint32_t)keys[i];
      const int c = cls_b[anchor];
      mycls[u] = c;
      C.lcls[i] = (unsigned short)c;
      const int r = atomicAdd(&C.ccount[c], 1);
      C.slotof[i] = (unsigned char)min(r, 255);
    }
  }
  __syncthreads();
  if (wave == 0) {   // class starts (ncp <= 128: two classes per lane) and the largest class
    const int a = 2 * lane < ncp ? C.ccount[2 * lane] : 0, b2 = 2 * lane + 1 < ncp ? C.ccount[2 * lane + 1] : 0;
    int incl = a + b2, big = max(a, b2);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int u = __shfl_up(incl, d), g = __shfl_xor(big, d);
      if (lane >= d) incl += u;
      big = max(big, g);
    }
    if (2 * lane < ncp) C.cstart[2 * lane] = incl - a - b2;
    if (2 * lane + 1 < ncp) C.cstart[2 * lane + 1] = incl - b2;
    if (lane == 0) C.misc[0] = big;
  }
  __syncthreads();
  if (C.misc[0] > 64 * kCpWords) return false;   // uniform
  // The pair tests below cost one serial sweep over the LARGEST class per lane (~0.18 us per member measured), the
  // sorted sweep of the fallback ~0.05 us per candidate of the image: models with a handful of classes (2-5 classes,
  // 100-250 candidates each) are served faster by the fallback (nc = 3, 425 candidates, largest class 211: 59 us
  // here against 33 us there), scenes spread over many classes by this path.  Same keeps either way.
  if (C.misc[0] > max(48, (3 * n) / 10 - 50)) return false;   // uniform
  const int W = (C.misc[0] + 63) >> 6;             // words in use for this image

  // grouped keys and class-offset boxes; x-extent of the candidates before the offset
  float lo_x = INFINITY, hi_x = -INFINITY;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int i = tid + kSortThreads * u;
    if (i < n) {
      const uint64_t key = keys[i];
      const uint32_t anchor = ~(uint32_t)key;
      const int c = mycls[u];
      const int g = C.cstart[c] + C.slotof[i];
      const float cx = ldf(pb, anchor), cy = ldf(pb, (size_t)A + anchor), w = ldf(pb, (size_t)2 * A + anchor), h = ldf(pb, (size_t)3 * A + anchor);
      const float hw = w / 2.0f, hh = h / 2.0f;
      const float off = (float)c * max_wh;
      gkey[g] = key;
      gbox[g] = make_float4((cx - hw) + off, (cy - hh) + off, (cx + hw) + off, (cy + hh) + off);
      C.gcls[g] = (unsigned short)c;
      // NaN coordinates must fail the extent test below: fminf / fmaxf would drop them
      lo_x = (cx - hw) < lo_x ? (cx - hw) : ((cx - hw) >= lo_x ? lo_x : NAN);
      hi_x = (cx + hw) > hi_x ? (cx + hw) : ((cx + hw) <= hi_x ? hi_x : NAN);
    }
  }
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) {
    const float l2 = __shfl_xor(lo_x, d), h2 = __shfl_xor(hi_x, d);
    lo_x = (l2 < lo_x) ? l2 : ((l2 >= lo_x) ? lo_x : NAN);
    hi_x = (h2 > hi_x) ? h2 : ((h2 <= hi_x) ? hi_x : NAN);
  }
  if (lane == 0) { C.ext[wave] = lo_x; C.ext[16 + wave] = hi_x; }
  for (int e = tid; e < ncp * kCpWords; e += kSortThreads) {
    const int m = C.ccount[e / kCpWords] - 64 * (e % kCpWords);   // members in this word
    C.keep_a[e] = m >= 64 ? ~0ull : (m > 0 ? ((1ull << m) - 1ull) : 0ull);   // generation 0: everybody
    C.keep_b[e] = 0ull;
  }
  __syncthreads();
  {
    float lo = C.ext[0], hi = C.ext[16];
    for (int w = 1; w < NW; ++w) {
      const float l2 = C.ext[w], h2 = C.ext[16 + w];
      lo = (l2 < lo) ? l2 : ((l2 >= lo) ? lo : NAN);
      hi = (h2 > hi) ? h2 : ((h2 <= hi) ? hi : NAN);
    }
    // boxes of different classes cannot intersect when every candidate lies inside one max_wh-wide band
    if (!((hi - lo) <= max_wh - 1.0f)) return false;   // uniform; false for NaN.  keys[0..n) are untouched
  }

  // 2. one lane per candidate, in class-grouped order (the lanes of a wave walk classes of similar length and read
  //    the same members: LDS broadcasts): bit r = member r of my class has a larger key and would suppress me
  uint64_t bits[2][kCpWords] = {};
  int myc[2] = {0, 0}, myslot[2] = {0, 0};
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int g = tid + kSortThreads * u;
    if (g >= n) break;
    const int c = C.gcls[g], sc = C.cstart[c], mc = C.ccount[c];
    myc[u] = c; myslot[u] = g - sc;
    const uint64_t mykey = gkey[g];
    const float4 cb = gbox[g];
    const float carea = (cb.z - cb.x) * (cb.w - cb.y);
#pragma unroll
    for (int w = 0; w < kCpWords; ++w) {
      if (w >= W) break;   // uniform
      uint64_t bset = 0ull;
      const int r1 = min(64, mc - 64 * w);
      for (int r = 0; r < r1; ++r) {
        const uint64_t kr = gkey[sc + 64 * w + r];
        const float4 kb = gbox[sc + 64 * w + r];
        if (kr > mykey && iou_gt(kb.x, kb.y, kb.z, kb.w, (kb.z - kb.x) * (kb.w - kb.y), cb.x, cb.y, cb.z, cb.w, carea, thr)) bset |= 1ull << r;
      }
      bits[u][w] = bset;
    }
  }

  // 3. fixpoint rounds over all classes at once
  uint64_t *cur = C.keep_a, *nxt = C.keep_b;
  for (int round = 0; round < 64 * kCpWords + 1; ++round) {   // a class of m members settles within m rounds
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int g = tid + kSortThreads * u;
      if (g < n) {
        uint64_t suppressed = 0ull;
#pragma unroll
        for (int w = 0; w < kCpWords; ++w)
          if (w < W) suppressed |= bits[u][w] & cur[myc[u] * kCpWords + w];
        if (suppressed == 0ull)
          atomicOr(reinterpret_cast<unsigned long long *>(&nxt[myc[u] * kCpWords + (myslot[u] >> 6)]), 1ull << (myslot[u] & 63));
      }
    }
    __syncthreads();
    int changed = 0;
    for (int e = tid; e < ncp * kCpWords; e += kSortThreads) {
      changed |= nxt[e] != cur[e];
      cur[e] = 0ull;   // becomes the next round's accumulator
    }
    changed = __syncthreads_or(changed);
    uint64_t *t = cur; cur = nxt; nxt = t;
    if (!changed) break;
  }
  // cur = the kept members of every class

  // 4. the kept keys, compacted into keys[0..Kt) (the list itself is no longer needed: gkey holds every key), then
  //    sorted: the output order and the max_det cut
  bool kept[2];
  int mine = 0;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int g = 2 * tid + u;   // thread t owns grouped positions 2t, 2t + 1
    kept[u] = false;
    if (g < n) {
      const int c = C.gcls[g], sl = g - C.cstart[c];
      kept[u] = (cur[c * kCpWords + (sl >> 6)] >> (sl & 63)) & 1ull;
    }
    mine += kept[u] ? 1 : 0;
  }
  int incl = mine;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int u = __shfl_up(incl, d);
    if (lane >= d) incl += u;
  }
  if (lane == 63) C.misc[1 + wave] = incl;
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < NW; ++w) {
    const int v = C.misc[1 + w];
    if (w < wave) before += v;
    total += v;
  }
  int idx = before + incl - mine;
#pragma unroll
  for (int u = 0; u < 2; ++u)
    if (kept[u]) keys[idx++] = gkey[2 * tid + u];
  const int PK = next_pow2(max(total, 1));
  for (int i = total + tid; i < PK; i += kSortThreads) keys[i] = 0ull;
  __syncthreads();
  bitonic_desc_lds(keys, PK);
  if (tid == 0) *sK = min(total, max_det);
  __syncthreads();
  return true;
}

struct FinalizeParams {
  float *det;
  const int *count;
  int batch, max_det, det_stride, nc;
  int model_h, model_w;
  const int *orig_hw;
  int uh0, uw0;
  int pad_round;
  const double *thres;
  double uniform_thres;
  uint8_t *keep;
  int *keep_count;
  int *mask_slot;
  int *slot_map;
  int *slot_base;
  int kpt_col0, kpt_n, kpt_ndim;   // fused form only: keypoint columns of the rows (kpt_n = 0: none)
};

// Python round(): nearest, ties to even
__device__ __forceinline__ double py_round(double x) { return rint(x); }

// ops.scale_boxes: gain / pad are Python floats (fp64), applied to the fp32 tensor with IEEE division
struct BoxScale {
  float px, py, gain, w0, h0;
};
__device__ __forceinline__ BoxScale box_scale(const FinalizeParams &F, int b) {
  const int h0 = F.orig_hw ? F.orig_hw[2 * b] : F.uh0;
  const int w0 = F.orig_hw ? F.orig_hw[2 * b + 1] : F.uw0;
  const double gh = (double)F.model_h / (double)h0, gw = (double)F.model_w / (double)w0;
  const double gain = gh < gw ? gh : gw;
  double px = ((double)F.model_w - (double)w0 * gain) / 2.0;
  double py = ((double)F.model_h - (double)h0 * gain) / 2.0;
  if (F.pad_round) { px = py_round(px - 0.1); py = py_round(py - 0.1); }
  return BoxScale{(float)px, (float)py, (float)gain, (float)w0, (float)h0};
}
__device__ __forceinline__ float scale_x(const BoxScale &s, float x) { return fminf(fmaxf((x - s.px) / s.gain, 0.0f), s.w0); }
__device__ __forceinline__ float scale_y(const BoxScale &s, float y) { return fminf(fmaxf((y - s.py) / s.gain, 0.0f), s.h0); }

struct SortNmsParams {
  const void *pred;   // fp32 or fp16 head, see the kernels' InT
  int fin;            // != 0: finalize (F) in the epilogue -- lmi_nms_v8_finalize
  FinalizeParams F;
  int nc, nm, A;
  float iou_thr;  // effective fp32 threshold (see host side)
  int agnostic, max_det, max_nms;
  float max_wh;
  uint64_t *keys;
  int P;
  const int *cls_of;
  float4 *sbox;
  const int *blk_count;   // [batch][nblk] candidates per filter workgroup
  int nblk, seg;
  int scr_off;            // byte offset of the gather's LDS scratch
  float *out_det;
  int *out_count;
  int *out_anchor;
  int lds_keys;  // capacity (keys) of the LDS sort buffer
  int lbox_off;  // byte offset of the LDS box buffer (lds_keys float4)
  float2 *scd;   // rotated: covariance c and reach per sorted candidate (global copy)
  int rot_groups; // rotated: workgroups per image that share the pair tests (1: one workgroup does everything)
  int pair_groups; // axis-aligned: workgroups per image of nms_pair_lists (0: that kernel did not run)
  int *pair_flags; // [batch] list form: != 0: the image is left to the sorted sweep
};

// Fused per-image half of `lmi_finalize_detections` (finalize_boxes_kernel is the stand-alone form): the per-class
// threshold and the ranks of image b's K kept rows (the boxes were scaled while the rows were written).  The prefix
// over the images that turns ranks into dense global mask slots stays a launch of its own (finalize_slots_kernel):
// doing it here in "whichever workgroup finishes last" needs device-scope fences, and on this multi-XCD part those
// write back and invalidate the L2s -- measured: the fused call went from 72 to 93 us and the letterbox running
// beside it from 100 to 158 us.  All kSortThreads lanes call this.  kpos / keys may live at the start of the LDS,
// where `lds` points: they are read into registers before `lds` is written.
__device__ void finalize_epilogue(const SortNmsParams &S, int b, int K, const int *kpos, const uint64_t *keys, int *lds) {
  const FinalizeParams &F = S.F;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int NW = kSortThreads / 64;
  bool k = false;
  if (tid < K) {
    const uint64_t key = keys[kpos[tid]];
    const uint32_t anchor = ~(uint32_t)key;
    const float conf = sortable2f((uint32_t)(key >> 32));
    const int c = S.cls_of[(size_t)b * S.A + anchor];
    const double t = F.thres ? ((c >= 0 && c < F.nc) ? F.thres[c] : 1.0) : F.uniform_thres;
    k = (double)conf > t;
  }
  __syncthreads();   // kpos / keys (possibly at the start of the LDS) are in registers now
  const uint64_t mask = __ballot(k);
  if (lane == 0) lds[wave] = __builtin_popcountll(mask);
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < NW; ++w) {
    const int t = lds[w];
    if (w < wave) before += t;
    total += t;
  }
  if (tid < F.max_det) {
    F.keep[(size_t)b * F.max_det + tid] = k ? 1 : 0;
    if (F.mask_slot) F.mask_slot[(size_t)b * F.max_det + tid] = k ? before + __builtin_popcountll(mask & ((1ull << lane) - 1ull)) : -1;
  }
  if (tid == 0) F.keep_count[b] = total;
}

// The K kept candidates of image b (keys[kpos[t]], t < K, in score order) -> rows x1,y1,x2,y2 (no class offset), conf,
// cls, coef..., their anchors and the count; with S.fin the boxes leave in original-image space (fused `ops.scale_boxes`
// + clip) and the per-class threshold / ranks follow (finalize_epilogue).  All kSortThreads lanes call this.
template <typename InT>
__device__ __forceinline__ void emit_rows(const SortNmsParams &S, int b, int K, const int *kpos, const uint64_t *keys, const InT *pb,
                                          unsigned char *smem) {
  const int tid = threadIdx.x;
  const int A = S.A;
  if (tid == 0) S.out_count[b] = K;
  const int row = 6 + S.nm;
  float *od = S.out_det + (size_t)b * S.max_det * row;
  BoxScale bs{};
  float kpx = 0.0f, kpy = 0.0f;   // ops.scale_coords on the keypoint columns: the pads are NOT rounded, unlike scale_boxes'
  const int k0 = S.fin ? S.F.kpt_col0 : 0, k1 = (S.fin && S.F.kpt_n > 0) ? k0 + S.F.kpt_n * S.F.kpt_ndim : 0;
  if (S.fin) {
    bs = box_scale(S.F, b);
    if (k1 > 0) {
      const int h0 = S.F.orig_hw ? S.F.orig_hw[2 * b] : S.F.uh0, w0 = S.F.orig_hw ? S.F.orig_hw[2 * b + 1] : S.F.uw0;
      const double gh = (double)S.F.model_h / (double)h0, gw = (double)S.F.model_w / (double)w0;
      const double gain = gh < gw ? gh : gw;
      kpx = (float)(((double)S.F.model_w - (double)w0 * gain) / 2.0);
      kpy = (float)(((double)S.F.model_h - (double)h0 * gain) / 2.0);
    }
  }
  for (int e = tid; e < K * row; e += kSortThreads) {
    const int t = e / row, col = e - t * row;
    const uint64_t key = keys[kpos[t]];
    const uint32_t anchor = ~(uint32_t)key;
    float v;
    if (col < 4) {
      const float c0 = ldf(pb, (size_t)(col & 1) * A + anchor);
      const float half = ldf(pb, (size_t)(2 + (col & 1)) * A + anchor) / 2.0f;
      v = (col < 2) ? (c0 - half) : (c0 + half);
      if (S.fin) v = (col & 1) ? scale_y(bs, v) : scale_x(bs, v);
    } else if (col == 4) {
      v = sortable2f((uint32_t)(key >> 32));
    } else if (col == 5) {
      v = (float)S.cls_of[(size_t)b * A + anchor];
    } else {
      v = ldf(pb, (size_t)(4 + S.nc + (col - 6)) * A + anchor);
      if (col >= k0 && col < k1) {   // scale_coords_kernel's arithmetic: (p - pad) / gain, clipped to the original frame
        const int d = (col - k0) % S.F.kpt_ndim;
        if (d == 0) v = fminf(fmaxf((v - kpx) / bs.gain, 0.0f), bs.w0);
        else if (d == 1) v = fminf(fmaxf((v - kpy) / bs.gain, 0.0f), bs.h0);
      }
    }
    od[e] = v;
    if (col == 0) S.out_anchor[(size_t)b * S.max_det + t] = (int)anchor;
  }
  NMS_STAMP(6);
  if (S.fin) {
    __syncthreads();   // every reader of the sort buffers is done: the epilogue reuses the start of the LDS
    finalize_epilogue(S, b, K, kpos, keys, reinterpret_cast<int *>(smem));
  }
}

// ---------------------------------------------------------------------------------------------
// Suppressor lists (round 3): the greedy pass of torchvision's nms is the unique fixpoint of
//     keep_j  =  no kept i with a LARGER key has IoU(i, j) > thr,
// so WHO could suppress j -- the pairs (i, j), key_i > key_j, IoU > thr -- can be found for all j independently, by as
// many workgroups as the launch leaves CUs for, before anybody knows who is kept.  That is what one CU cannot do fast
// for agnostic NMS (all candidates in one class: the sorted sweep's window-vs-kept tests, ~110 k pair tests for a
// dense image, are bound by the vector ALU of the ONE CU the image's workgroup runs on: 65 us for one image) and what
// the class-parallel path does inside a workgroup for classes of <= 256 members.  nms_pair_lists: G workgroups per
// image (host: CUs / batch, at most 64, at least kPairMinGroups -- below that the sweep, which tests against kept boxes
// only, is cheaper than all n^2 pairs).  Every workgroup gathers the image's candidates in the filter's order (no
// sort), decodes their (class-offset) boxes into LDS and owns every (16 G)-th column j per wave: lanes hold 64
// candidates i, the wave's columns come by LDS broadcast, a cheap exact filter (larger key, x and y extents meet: both
// necessary for IoU > 0) queues the near pairs in LDS, and the queue is evaluated 64 pairs at a time with every lane on
// the IoU formula.  Hits go to column j's list in global memory -- up to kPairL suppressor indices (positions in the
// gathered list) and the hit count.  nms_sort_suppress then gathers the same list in the same order, loads the lists
// and resolves the fixpoint with per-candidate states (kept / dropped / open) in a few rounds, compacts the kept keys and
// sorts only those.  Images with more candidates than pair_cap() or max_nms, or with a list longer than kPairL, take the
// sorted sweep as before (nms_pair_lists leaves such an image after counting its candidates).
constexpr int kPairL = 48;           // suppressors listed per candidate (a multiple of 8: whole 16-byte loads)
constexpr int kPairCapMax = 2048;    // candidates per image
constexpr int kPairQueue = 128;      // near pairs a wave queues before it evaluates 64 of them
constexpr int kPairMinGroups = 16;
constexpr int kPairMaxGroups = 64;
// The lists live in the sorted-box and covariance scratch taken together (24 bytes per anchor, contiguous over the batch):
// image b's area starts 24 A b bytes into it and holds cap lists of kPairL u16, then cap hit counts.  The sweep only uses
// that scratch for images above its LDS buffers, in the launch AFTER nms_resolve_lists has consumed the lists.
__host__ __device__ inline int pair_cap(int anchors) {
  const long c = ((long)anchors * 24 / (2 * kPairL + 4)) & ~7L;
  return c < kPairCapMax ? (int)c : kPairCapMax;
}
__device__ __forceinline__ unsigned char *pair_area(const SortNmsParams &S, int b) {
  return reinterpret_cast<unsigned char *>(S.sbox) + (size_t)b * 24 * S.A;
}
constexpr size_t resolve_lds_bytes(int p2cap, int max_det, int cap) {
  return (size_t)p2cap * 8 + (size_t)((max_det + 3) & ~3) * 4 + (size_t)((cap + 15) & ~15) + 32 * 4 + kGatherLds;
}
constexpr size_t pair_lds_bytes(int p2cap) { return (size_t)p2cap * (8 + 16 + 4) + (size_t)(kSortThreads / 64) * kPairQueue * 4 + kGatherLds; }

template <typename InT>
__global__ __launch_bounds__(kSortThreads) void nms_pair_lists(SortNmsParams S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NW = kSortThreads / 64;
  const int G = S.pair_groups;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = S.A, C = 4 + S.nc + S.nm;
  const int cap = pair_cap(A), p2cap = next_pow2(cap);
  uint64_t *lkeys = reinterpret_cast<uint64_t *>(smem);
  float4 *lbox = reinterpret_cast<float4 *>(smem + (size_t)p2cap * 8);
  int *lcnt = reinterpret_cast<int *>(smem + (size_t)p2cap * 24);
  uint32_t *queue = reinterpret_cast<uint32_t *>(smem + (size_t)p2cap * 28) + wave * kPairQueue;
  int *scr = reinterpret_cast<int *>(smem + (size_t)p2cap * 28 + (size_t)NW * kPairQueue * 4);
  const InT *pb = reinterpret_cast<const InT *>(S.pred) + (size_t)b * C * A;

  int P2 = 0, n_total = 0;
  bool in_lds = true;
  NMS_STAMP(10);
  const int n = gather_candidates(S.blk_count + (size_t)b * S.nblk, S.nblk, S.seg, A, p2cap, lkeys, S.keys + (size_t)b * S.P, nullptr, scr,
                                  P2, in_lds, nullptr, n_total, min(cap, S.max_nms));
  if (n <= 0) return;   // no candidates, or more than the lists serve: the sorted sweep takes this image
  NMS_STAMP(11);
  for (int i = tid; i < n; i += kSortThreads) {
    const uint32_t anchor = ~(uint32_t)lkeys[i];
    const float cx = ldf(pb, anchor), cy = ldf(pb, (size_t)A + anchor), w = ldf(pb, (size_t)2 * A + anchor), h = ldf(pb, (size_t)3 * A + anchor);
    const float hw = w / 2.0f, hh = h / 2.0f;
    const float c = S.agnostic ? 0.0f : (float)S.cls_of[(size_t)b * A + anchor] * S.max_wh;
    lbox[i] = make_float4((cx - hw) + c, (cy - hh) + c, (cx + hw) + c, (cy + hh) + c);   // as the sweep decodes them
    lcnt[i] = 0;
  }
  __syncthreads();
  NMS_STAMP(12);

  uint16_t *plist = reinterpret_cast<uint16_t *>(pair_area(S, b));
  int *pcnt = reinterpret_cast<int *>(pair_area(S, b) + (size_t)cap * kPairL * 2);
  const int nw = NW * G, w0 = g * NW + wave;   // this wave's columns: w0, w0 + nw, ...
  int qn = 0;                                  // queued pairs (wave-uniform)
  auto evaluate = [&](int cnt) {
    if (lane < cnt) {
      const uint32_t pr = queue[lane];
      const int i = (int)(pr & 0xffffu), j = (int)(pr >> 16);
      const float4 bi = lbox[i], bj = lbox[j];
      if (iou_gt(bi.x, bi.y, bi.z, bi.w, (bi.z - bi.x) * (bi.w - bi.y), bj.x, bj.y, bj.z, bj.w, (bj.z - bj.x) * (bj.w - bj.y), S.iou_thr)) {
        const int slot = atomicAdd(&lcnt[j], 1);   // column j belongs to this wave alone
        if (slot < kPairL) plist[(size_t)j * kPairL + slot] = (uint16_t)i;
      }
    }
  };
  // Cheap test: i has the larger key and the boxes' extents meet strictly in x and in y -- necessary for inter > 0 in
  // iou_gt (min(x2) > max(x1) implies both cross comparisons; a NaN coordinate makes IoU NaN and the formula says "no"
  // whatever this test says).  Plain comparisons joined by & (no short-circuit: the operands are in registers).  (An
  // area-ratio test on top -- IoU <= min(area) / max(area) -- was tried: fewer pairs reach the formula, but the loop is
  // bound by the cheap tests, not by the formula: no gain.)
  auto push = [&](bool near, uint64_t m, int i, int j) {
    if (near) queue[qn + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)i | ((uint32_t)j << 16);
    qn += __builtin_popcountll(m);
    __builtin_amdgcn_wave_barrier();
    if (qn >= 64) {
      evaluate(64);
      __builtin_amdgcn_wave_barrier();
      const uint32_t t = lane < qn - 64 ? queue[64 + lane] : 0u;
      __builtin_amdgcn_wave_barrier();
      if (lane < qn - 64) queue[lane] = t;
      __builtin_amdgcn_wave_barrier();
      qn -= 64;
    }
  };
  // Columns in pairs held in registers (a wave has n / (16 G) of them: one or two with 64 workgroups per image), the
  // candidates i streamed past them 128 at a time: the only LDS reads in the loop are the lanes' own (key, box), and
  // those of the next block do not depend on this block's tests.
  const int ncol = n > w0 ? (n - w0 + nw - 1) / nw : 0;
  for (int c = 0; c < ncol; c += 2) {
    const int j0 = w0 + c * nw, j1 = j0 + nw;
    const bool two = j1 < n;
    const uint64_t kj0 = lkeys[j0], kj1 = two ? lkeys[j1] : ~0ull;   // ~0: no key is larger, the second column never hits
    const float4 bj0 = lbox[j0], bj1 = lbox[two ? j1 : j0];
    for (int i0 = 0; i0 < n; i0 += 128) {
      const int ia = i0 + lane, ib = ia + 64;
      const uint64_t ka = ia < n ? lkeys[ia] : 0ull, kb = ib < n ? lkeys[ib] : 0ull;   // 0: below every key
      const float4 ba = lbox[min(ia, n - 1)], bb = lbox[min(ib, n - 1)];
      const bool a0 = (ka > kj0) & (ba.z > bj0.x) & (bj0.z > ba.x) & (ba.w > bj0.y) & (bj0.w > ba.y);
      const bool a1 = (ka > kj1) & (ba.z > bj1.x) & (bj1.z > ba.x) & (ba.w > bj1.y) & (bj1.w > ba.y);
      const bool b0 = (kb > kj0) & (bb.z > bj0.x) & (bj0.z > bb.x) & (bb.w > bj0.y) & (bj0.w > bb.y);
      const bool b1 = (kb > kj1) & (bb.z > bj1.x) & (bj1.z > bb.x) & (bb.w > bj1.y) & (bj1.w > bb.y);
      const uint64_t ma0 = __ballot(a0), ma1 = __ballot(a1), mb0 = __ballot(b0), mb1 = __ballot(b1);
      if ((ma0 | ma1 | mb0 | mb1) == 0ull) continue;
      if (ma0) push(a0, ma0, ia, j0);
      if (ma1) push(a1, ma1, ia, j1);
      if (mb0) push(b0, mb0, ib, j0);
      if (mb1) push(b1, mb1, ib, j1);
    }
  }
  NMS_STAMP(13);
  __builtin_amdgcn_wave_barrier();
  evaluate(qn);
  __builtin_amdgcn_wave_barrier();
  for (int j = w0 + nw * lane; j < n; j += nw * 64) pcnt[j] = lcnt[j];
  NMS_STAMP(14);
}

// Position (0..255) of the t-th set bit (t counted from 0, t < number of set bits) of the 256-bit set am[0..3];
// ac[q] = popcount(am[q]).
__device__ __forceinline__ int nth_alive(const uint64_t (&am)[4], const int (&ac)[4], int t) {
  int q = 0;
  uint64_t w = am[0];
  if (t >= ac[0]) {
    t -= ac[0]; q = 1; w = am[1];
    if (t >= ac[1]) {
      t -= ac[1]; q = 2; w = am[2];
      if (t >= ac[2]) { t -= ac[2]; q = 3; w = am[3]; }
    }
  }
  int pos = 0;
  uint32_t v = (uint32_t)w;
  int c = __builtin_popcount(v);
  if (t >= c) { t -= c; pos = 32; v = (uint32_t)(w >> 32); }
  c = __builtin_popcount(v & 0xffffu); if (t >= c) { t -= c; pos += 16; v >>= 16; }
  c = __builtin_popcount(v & 0xffu);   if (t >= c) { t -= c; pos += 8; v >>= 8; }
  c = __builtin_popcount(v & 0xfu);    if (t >= c) { t -= c; pos += 4; v >>= 4; }
  c = __builtin_popcount(v & 0x3u);    if (t >= c) { t -= c; pos += 2; v >>= 2; }
  if (t >= (int)(v & 1u)) pos += 1;
  return 64 * q + pos;
}

// Second half of the list form (see nms_pair_lists): one workgroup per image gathers the candidates again (same order),
// loads their suppressor lists and resolves the greedy fixpoint with per-candidate states; the kept keys are compacted,
// sorted (only they) and written out like nms_sort_suppress does.  A kernel of its own, lean: inside nms_sort_suppress the
// two lists of kPairL indices per lane cost that kernel's other paths registers and spills (its candidate gather alone
// went from 5 to 11 us).  Images it cannot serve -- more candidates than the lists hold, or a list longer than kPairL --
// are flagged for the launch of nms_sort_suppress that follows (which leaves every other image at once).
// LDS: lds_keys keys | max_det positions | cap state bytes + 16 wave totals | the gather's scratch.
__device__ __forceinline__ int *pair_flag(const SortNmsParams &S, int b) { return S.pair_flags + b; }

template <typename InT>
__global__ __launch_bounds__(kSortThreads) void nms_resolve_lists(SortNmsParams S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = S.A, C = 4 + S.nc + S.nm;
  const int cap = pair_cap(A), p2cap = next_pow2(cap);
  uint64_t *lkeys = reinterpret_cast<uint64_t *>(smem);
  int *kpos = reinterpret_cast<int *>(smem + (size_t)p2cap * 8);
  unsigned char *state = reinterpret_cast<unsigned char *>(kpos + ((S.max_det + 3) & ~3));   // 0 open, 1 kept, 2 dropped
  int *misc = reinterpret_cast<int *>(state + ((cap + 15) & ~15));                             // wave totals of the compaction
  int *scr = misc + 32;
  const InT *pb = reinterpret_cast<const InT *>(S.pred) + (size_t)b * C * A;
  const uint16_t *plist = reinterpret_cast<const uint16_t *>(pair_area(S, b));
  const int *pcnt = reinterpret_cast<const int *>(pair_area(S, b) + (size_t)cap * kPairL * 2);

  NMS_STAMP(0);
  int P2 = 0, n_total = 0;
  bool in_lds = true;
  const int n = gather_candidates(S.blk_count + (size_t)b * S.nblk, S.nblk, S.seg, A, p2cap, lkeys, S.keys + (size_t)b * S.P, nullptr, scr,
                                  P2, in_lds, nullptr, n_total, min(cap, S.max_nms));
  NMS_STAMP(1);
  if (n < 0) {            // nms_pair_lists left this image too
    if (tid == 0) pair_flag(S, b)[0] = 1;
    return;
  }
  int K = 0;
  if (n > 0) {
      constexpr int NW = kSortThreads / 64;
      int cnt[2] = {0, 0}, st[2] = {3, 3};   // 3: no candidate
      uint64_t mykey[2] = {0ull, 0ull};
      uint4 lst[2][kPairL / 8];
      int over = 0;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int j = tid + kSortThreads * u;
        if (j < n) {
          cnt[u] = pcnt[j];
          over |= cnt[u] > kPairL ? 1 : 0;
          mykey[u] = lkeys[j];
          const uint4 *lp = reinterpret_cast<const uint4 *>(plist + (size_t)j * kPairL);
#pragma unroll
          for (int w4 = 0; w4 < kPairL / 8; ++w4)
            if (8 * w4 < cnt[u] && cnt[u] <= kPairL) lst[u][w4] = lp[w4];
          st[u] = cnt[u] == 0 ? 1 : 0;
          state[j] = (unsigned char)st[u];
        }
      }
      over = __syncthreads_or(over);   // also publishes the states
      NMS_STAMP(20);
      if (over) NMS_STAMP(25);
      if (!over) {
        // A candidate without suppressors is kept; one with a kept suppressor is dropped; one whose suppressors are all
        // dropped is kept.  A state changes once, from open to its final value, and every open candidate only waits for
        // candidates with larger keys, so the open candidate with the largest key can always be settled: no barrier is
        // needed for progress.  Every wave looks at its own open candidates again and again -- the others' states are
        // read from LDS as they are at that moment: a final state, or "open" (then the lane looks again) -- until its own
        // are settled; the workgroup meets once, afterwards.  (With a workgroup barrier per round: 7-9 rounds, 8 us for a
        // dense image.)  The bound only guards against a stuck wave; whoever is still open then sends the image to the
        // sweep.
        for (int it = 0; it < 4096; ++it) {
          if (__ballot(st[0] == 0 || st[1] == 0) == 0ull) break;   // this wave's candidates are settled
          bool progress = false;
#pragma unroll
          for (int u = 0; u < 2; ++u) {
            const bool open = st[u] == 0;
            bool any_kept = false, all_dropped = true;
#pragma unroll
            for (int w4 = 0; w4 < kPairL / 8; ++w4) {
              if (__ballot(open && 8 * w4 < cnt[u]) == 0ull) break;   // uniform: nobody in the wave has that many
              const uint32_t ws[4] = {lst[u][w4].x, lst[u][w4].y, lst[u][w4].z, lst[u][w4].w};
#pragma unroll
              for (int e = 0; e < 8; ++e) {
                if (open && 8 * w4 + e < cnt[u]) {
                  const int sv = state[(ws[e >> 1] >> (16 * (e & 1))) & 0xffffu];
                  any_kept |= sv == 1;
                  all_dropped &= sv == 2;
                }
              }
            }
            if (open && (any_kept || all_dropped)) {
              st[u] = any_kept ? 2 : 1;
              state[tid + kSortThreads * u] = (unsigned char)st[u];
              progress = true;
            }
          }
          __asm__ volatile("" ::: "memory");                            // the states are read afresh next time
          if (__ballot(progress) == 0ull) __builtin_amdgcn_s_sleep(2);  // waiting for other waves' candidates
        }
        if (__syncthreads_or(st[0] == 0 || st[1] == 0)) {               // (cannot happen short of the bound)
          if (tid == 0) pair_flag(S, b)[0] = 1;
          return;
        }
        NMS_STAMP(21);
        // the kept keys, compacted to the head of the list (every lane holds its candidates' keys), then sorted: the
        // output order and the S.max_det cut
        const int mine = (st[0] == 1 ? 1 : 0) + (st[1] == 1 ? 1 : 0);
        int incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int t = __shfl_up(incl, d);
          if (lane >= d) incl += t;
        }
        if (lane == 63) misc[wave] = incl;
        __syncthreads();
        int before = 0, total = 0;
        for (int w = 0; w < NW; ++w) {
          const int t = misc[w];
          if (w < wave) before += t;
          total += t;
        }
        int idx = before + incl - mine;
#pragma unroll
        for (int u = 0; u < 2; ++u)
          if (st[u] == 1) lkeys[idx++] = mykey[u];
        const int PK = next_pow2(max(total, 1));
        for (int i = total + tid; i < PK; i += kSortThreads) lkeys[i] = 0ull;
        __syncthreads();
        bitonic_desc_lds(lkeys, PK);
        K = min(total, S.max_det);
        for (int t = tid; t < K; t += kSortThreads) kpos[t] = t;   // the kept keys lie sorted at the head of the list
        __syncthreads();
        NMS_STAMP(22);
      } else {
        if (tid == 0) pair_flag(S, b)[0] = 1;   // a list ran over: the sorted sweep
        return;
      }
  }
  if (tid == 0) pair_flag(S, b)[0] = 0;
  emit_rows<InT>(S, b, K, kpos, lkeys, pb, smem);
}

template <typename InT>
__global__ __launch_bounds__(kSortThreads) void nms_sort_suppress(SortNmsParams S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t *lkeys = reinterpret_cast<uint64_t *>(smem);
  const int md = (S.max_det + 1) & ~1;  // even stride keeps the 64-bit arrays behind it 8-byte aligned
  float *kx1 = reinterpret_cast<float *>(smem + (size_t)S.lds_keys * 8);
  float *ky1 = kx1 + md;
  float *kx2 = ky1 + md;
  float *ky2 = kx2 + md;
  float *karea = ky2 + md;
  int *kpos = reinterpret_cast<int *>(karea + md);
  uint64_t *pairmask = reinterpret_cast<uint64_t *>(kpos + md);
  int *supp = reinterpret_cast<int *>(pairmask + 64);
  int *sK = supp + kNmsWindow;
  // sorted class-offset boxes stay in LDS when the candidates fit the LDS sort buffer (no global round trip per chunk)
  float4 *lbox = reinterpret_cast<float4 *>(smem + S.lbox_off);

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = S.A, C = 4 + S.nc + S.nm;
  // behind nms_pair_lists + nms_resolve_lists: only the images those two could not serve (uniform over the workgroup)
  if (S.pair_groups > 0 && pair_flag(S, b)[0] == 0) return;
  const InT *pb = reinterpret_cast<const InT *>(S.pred) + (size_t)b * C * A;
  uint64_t *gkeys = S.keys + (size_t)b * S.P;
  float4 *sbox = S.sbox + (size_t)b * A;

  NMS_STAMP(0);
  // ---- the image's candidates, compact, into LDS (or the global list when they do not fit) ----------------------
  int P2 = 0, n_total = 0;
  bool in_lds = true;
  // more candidates than the LDS buffers hold: first only the head of the sorted list (top_keys; the box buffer is
  // free until the boxes are decoded and lends it its histogram)
  int n = gather_candidates(S.blk_count + (size_t)b * S.nblk, S.nblk, S.seg, A, S.lds_keys, lkeys, gkeys,
                            reinterpret_cast<uint64_t *>(sbox), reinterpret_cast<int *>(smem + S.scr_off), P2, in_lds,
                            reinterpret_cast<int *>(lbox), n_total);
  if (n == 0) {
    if (tid == 0) S.out_count[b] = 0;
    if (S.fin) finalize_epilogue(S, b, 0, nullptr, nullptr, reinterpret_cast<int *>(smem));
    return;
  }
  bool head_only = n < n_total;
  NMS_STAMP(1);
  // one pass over the candidates held in lkeys / gkeys (n, P2, in_lds): class-parallel or sorted sweep; returns K
  auto attempt = [&](const int n, const int P2, const bool in_lds) -> int {
    if (tid < kNmsWindow) supp[tid] = 0;
    if (tid == 0) *sK = 0;
    __syncthreads();

    int K = 0;
    bool done = false;
    // ---- class-parallel suppression on the UNSORTED candidates (see class_parallel_nms): per-class NMS, few classes,
    //      every candidate inside the max_nms cut, the candidates in the lower half of the LDS key buffer (the upper
    //      half takes the grouped copy) and room for the scratch behind the grouped boxes -------------------------------
    const int ncp = (S.nc + 1) & ~1;
    const bool cp_try = in_lds && !S.agnostic && S.nc <= kCpMaxClasses && P2 <= 2 * kSortThreads && n <= S.max_nms &&
                        2 * P2 <= S.lds_keys && (size_t)(S.lds_keys - P2) * sizeof(float4) >= cp_bytes(P2, ncp);
    if (cp_try) {
      const CpScratch CP = cp_carve(reinterpret_cast<unsigned char *>(lbox + P2), P2, ncp);
      done = class_parallel_nms(CP, lkeys, lkeys + P2, lbox, pb, S.cls_of + (size_t)b * A, A, n, P2, ncp, S.max_det, S.max_wh, S.iou_thr, sK);
      if (done) {
        K = *sK;
        for (int t = tid; t < K; t += kSortThreads) kpos[t] = t;   // the kept keys lie sorted at the head of the list
        __syncthreads();
      }
    }

    NMS_STAMP(2);
    // ---- otherwise: sort the whole list, decode the sorted candidates' (class-offset) boxes ----------------------------
    const int n_eff = min(n, S.max_nms);
    if (!done) {
      if (in_lds) bitonic_desc_lds(lkeys, P2);
      else bitonic_desc(gkeys, P2);
      NMS_STAMP(3);
      for (int i = tid; i < n_eff; i += kSortThreads) {
        const uint64_t key = in_lds ? lkeys[i] : gkeys[i];
        const uint32_t anchor = ~(uint32_t)key;
        const float cx = ldf(pb, anchor), cy = ldf(pb, (size_t)A + anchor), w = ldf(pb, (size_t)2 * A + anchor), h = ldf(pb, (size_t)3 * A + anchor);
        const float hw = w / 2.0f, hh = h / 2.0f;
        const float c = S.agnostic ? 0.0f : (float)S.cls_of[(size_t)b * A + anchor] * S.max_wh;
        const float4 bx = make_float4((cx - hw) + c, (cy - hh) + c, (cx + hw) + c, (cy + hh) + c);
        if (in_lds) lbox[i] = bx; else sbox[i] = bx;
      }
      __syncthreads();
    }

    // ---- greedy NMS over the sorted list (agnostic NMS, many classes, few but large classes, boxes wider than the
    //      class offset, or more candidates than the LDS buffers hold).  Every step looks at a WINDOW of the next 256
    //      candidates: (A) all of them are tested against the boxes kept so far -- whoever is hit is dead for good,
    //      kept boxes are final --, then the first 64 candidates still alive are compacted into one wave's lanes and
    //      (B) resolved among themselves exactly (pair masks + fixpoint).  The step consumes the window up to the 65th
    //      alive candidate (the whole window when at most 64 are alive).  In a dense scene nine candidates in ten die
    //      in (A) once the first boxes are kept, so the serial steps (each a chain of LDS round trips and three
    //      barriers) shrink from one per 64 candidates to one per 64 SURVIVORS of (A): 1152 candidates, agnostic:
    //      18 -> 7 steps. ----------------------------------------------------------------------------------------------
    NMS_STAMP(4);
    for (int base = 0; !done && base < n_eff && K < S.max_det;) {
      const int wn = min(kNmsWindow, n_eff - base);
      // A: every lane holds four candidates of the window (one per 64-candidate sub-chunk) and every wave walks every
      // sixteenth kept box, two per iteration so that the LDS broadcasts are issued together: a kept box read from
      // LDS serves four pair tests.  Most kept boxes cannot meet any candidate (another class -- the class offset
      // moves it 7680 px away -- or another place): a kept box whose x-extent meets no lane's (xx2 > xx1 is necessary
      // for inter > 0, exactly) is dropped by one wave-uniform branch before the areas and the division are touched.
      {
        float4 wb[4];
        float wa[4];
        bool hit[4] = {false, false, false, false};
  #pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int wp = 64 * q + lane;
          wb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
          if (wp < wn) wb[q] = in_lds ? lbox[base + wp] : sbox[base + wp];
          wa[q] = (wb[q].z - wb[q].x) * (wb[q].w - wb[q].y);
        }
        constexpr int NW = kSortThreads / 64;
        int k = wave;
        for (; k + NW < K; k += 2 * NW) {
          const float a0 = kx1[k], e0 = kx2[k], c0 = ky1[k], f0 = ky2[k], r0 = karea[k];
          const float a1 = kx1[k + NW], e1 = kx2[k + NW], c1 = ky1[k + NW], f1 = ky2[k + NW], r1 = karea[k + NW];
          bool o0 = false, o1 = false;
  #pragma unroll
          for (int q = 0; q < 4; ++q) {
            o0 = o0 | (fminf(e0, wb[q].z) > fmaxf(a0, wb[q].x));
            o1 = o1 | (fminf(e1, wb[q].z) > fmaxf(a1, wb[q].x));
          }
          if (__ballot(o0) != 0ull) {
  #pragma unroll
            for (int q = 0; q < 4; ++q) hit[q] = hit[q] | iou_gt(a0, c0, e0, f0, r0, wb[q].x, wb[q].y, wb[q].z, wb[q].w, wa[q], S.iou_thr);
          }
          if (__ballot(o1) != 0ull) {
  #pragma unroll
            for (int q = 0; q < 4; ++q) hit[q] = hit[q] | iou_gt(a1, c1, e1, f1, r1, wb[q].x, wb[q].y, wb[q].z, wb[q].w, wa[q], S.iou_thr);
          }
        }
        if (k < K) {
          const float a0 = kx1[k], e0 = kx2[k], c0 = ky1[k], f0 = ky2[k], r0 = karea[k];
  #pragma unroll
          for (int q = 0; q < 4; ++q) hit[q] = hit[q] | iou_gt(a0, c0, e0, f0, r0, wb[q].x, wb[q].y, wb[q].z, wb[q].w, wa[q], S.iou_thr);
        }
  #pragma unroll
        for (int q = 0; q < 4; ++q)
          if (hit[q] && 64 * q + lane < wn) supp[64 * q + lane] = 1;
      }
      __syncthreads();

      // the window's candidates still alive, as four 64-bit sets (every wave forms the same sets: no LDS round trip);
      // lane t takes the t-th of them
      uint64_t am[4];
      int ac[4];
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        am[q] = __ballot(64 * q + lane < wn && supp[64 * q + lane] == 0);
        ac[q] = __builtin_popcountll(am[q]);
      }
      const int alive_n = ac[0] + ac[1] + ac[2] + ac[3];
      const int m = min(64, alive_n);
      int mypos = 0;
      float4 cb = make_float4(0.f, 0.f, 0.f, 0.f);
      if (lane < m) {
        mypos = nth_alive(am, ac, lane);
        cb = in_lds ? lbox[base + mypos] : sbox[base + mypos];
      }
      const float carea = (cb.z - cb.x) * (cb.w - cb.y);

      // B: pair masks among the (up to) 64 alive candidates, by column: pairmask[j] = lanes i < j whose box would
      // suppress j's (same x-extent shortcut)
  #pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int j = __builtin_amdgcn_readfirstlane(wave) * 4 + q;  // wave-uniform: the box of lane j comes by readlane
        const float bx1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb.x), j));
        const float bx2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb.z), j));
        uint64_t mask = 0ull;
        if (__ballot((lane < j) && fminf(cb.z, bx2) > fmaxf(cb.x, bx1)) != 0ull) {
          const float by1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb.y), j));
          const float by2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cb.w), j));
          const float barea = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(carea), j));
          const bool p = (lane < j) && (j < m) && iou_gt(cb.x, cb.y, cb.z, cb.w, carea, bx1, by1, bx2, by2, barea, S.iou_thr);
          mask = __ballot(p);
        }
        if (lane == 0) pairmask[j] = mask;
      }
      __syncthreads();

      if (wave == 0) {
        // Greedy suppression among them as a fixpoint: S = candidates kept so far; lane j stays iff no kept i < j
        // suppresses it.  After t rounds the first t lanes are final (by induction on j), so the loop ends within 64
        // rounds and a round that changes nothing has reached the greedy answer -- in practice after the length of the
        // longest suppression chain, 2-5 rounds, instead of one serial step per kept box.
        const uint64_t alive = __ballot(lane < m);
        const uint64_t col = pairmask[lane];
        uint64_t keep_set = alive;
        for (int round = 0; round < 65; ++round) {   // 64 rounds settle every lane; the bound is belt and braces
          const uint64_t next = alive & ~__ballot((col & keep_set) != 0ull);
          if (next == keep_set) break;
          keep_set = next;
        }
        // at most max_det rows in all: the first (max_det - K) kept lanes of this step
        const int rank = __builtin_popcountll(keep_set & ((1ull << lane) - 1ull));
        const uint64_t keepm = __ballot(((keep_set >> lane) & 1ull) && K + rank < S.max_det);
        const int Kl = K + __builtin_popcountll(keepm);
        if ((keepm >> lane) & 1ull) {
          const int idx = K + __builtin_popcountll(keepm & ((1ull << lane) - 1ull));
          kx1[idx] = cb.x; ky1[idx] = cb.y; kx2[idx] = cb.z; ky2[idx] = cb.w;
          karea[idx] = carea;
          kpos[idx] = base + mypos;
        }
  #pragma unroll
        for (int q = 0; q < 4; ++q) supp[64 * q + lane] = 0;   // every wave read its flags before the barrier above
        // the step consumes the window up to the 65th alive candidate (>= 64 positions), all of it when at most 64
        // are alive (>= 1): found by wave 0 alone, the other waves read it with K
        const int used = alive_n > 64 ? nth_alive(am, ac, 64) : wn;
        if (lane == 0) { sK[0] = Kl; sK[1] = used; }
      }
      __syncthreads();
      K = sK[0];
      base += sK[1];
    }

    NMS_STAMP(5);
    return K;
  };
  int K = attempt(n, P2, in_lds);
  // the head of the list ended short of max_det keeps although candidates remain: the whole list after all (sorted in
  // global memory).  Uniform: every lane sees the same K and counts.
  if (head_only && K < S.max_det && n < min(n_total, S.max_nms)) {
    head_only = false;
    __syncthreads();
    n = gather_candidates(S.blk_count + (size_t)b * S.nblk, S.nblk, S.seg, A, S.lds_keys, lkeys, gkeys,
                          reinterpret_cast<uint64_t *>(sbox), reinterpret_cast<int *>(smem + S.scr_off), P2, in_lds, nullptr, n_total);
    K = attempt(n, P2, in_lds);
  }

  emit_rows<InT>(S, b, K, kpos, in_lds ? lkeys : gkeys, pb, smem);
}

// ---------------------------------------------------------------------------------------------
// Rotated boxes: `non_max_suppression(rotated=True)` -> `nms_rotated` (ultralytics 8.3.x; semantics and formula
// in oracle/obb.py).  Same sort as above, then NOT a greedy pass: candidate j (in score order) is kept iff
// probiou(i, j) < thr for EVERY higher-scored i, kept or not, so the columns are independent: lane j walks i < j
// and leaves at the first hit.  Per candidate the class-offset centre, the covariance terms a, b, c of its
// ellipse and max(ab - c^2, 0) are computed once.  Rows come out as [x, y, w, h, conf, cls, extras..., angle].
__device__ __forceinline__ float probiou(float x1, float y1, float a1, float b1, float c1, float d1, float x2, float y2,
                                         float a2, float b2, float c2, float d2) {
  const float eps = 1e-7f;
  const float sa = a1 + a2, sb = b1 + b2, sc = c1 + c2;
  const float det = sa * sb - sc * sc;
  const float den = det + eps;
  const float dy = y1 - y2, dx = x1 - x2;
  const float t1 = ((sa * (dy * dy) + sb * (dx * dx)) / den) * 0.25f;
  const float t2 = (((sc * (x2 - x1)) * (y1 - y2)) / den) * 0.5f;
  const float t3 = logf(det / (4.0f * sqrtf(d1 * d2) + eps) + eps) * 0.5f;
  float bd = t1 + t2 + t3;
  bd = bd != bd ? bd : fminf(fmaxf(bd, eps), 100.0f);   // torch.clamp keeps a NaN (then `NaN < thr` is false: a hit), fmaxf would not
  const float hd = sqrtf(1.0f - expf(-bd) + eps);
  return 1.0f - hd;
}

template <typename InT>
__global__ __launch_bounds__(kSortThreads) void nms_sort_rotated(SortNmsParams S) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t *lkeys = reinterpret_cast<uint64_t *>(smem);
  const int md = (S.max_det + 1) & ~1;
  int *kpos = reinterpret_cast<int *>(smem + (size_t)S.lds_keys * 8);
  int *wcnt = kpos + md;  // 16 wave counts + running total
  float4 *lbox = reinterpret_cast<float4 *>(smem + S.lbox_off);   // x + off, y + off, a, b
  float2 *lcd = reinterpret_cast<float2 *>(lbox + S.lds_keys);     // c, reach

  // G workgroups per image (host: as many as leave every workgroup a CU of its own).  The pair tests are independent
  // per column, unlike the greedy pass, so they can be shared out -- but one workgroup runs on ONE CU, and a dense
  // scene (300 large boxes of one class: 1.1 ms) is bound by its vector ALU.  Every workgroup of an image gathers,
  // sorts and decodes the SAME list in its own LDS (redundant, but on CUs that would idle, and without a
  // synchronisation between workgroups), tests only its own columns (wave w of group g: columns w + 16 g mod 16 G),
  // and leaves their keep flags -- group 0 also the sorted keys -- in the image's global scratch; nms_rotated_emit
  // (a second launch) compacts and writes the rows.  Images with more candidates than the LDS buffers hold are done by
  // group 0 alone, start to finish, as with G = 1.
  const int G = S.rot_groups > 1 ? S.rot_groups : 1;
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = S.A, C = 4 + S.nc + S.nm;
  const InT *pb = reinterpret_cast<const InT *>(S.pred) + (size_t)b * C * A;
  uint64_t *gkeys = S.keys + (size_t)b * S.P;
  int *gn = reinterpret_cast<int *>(S.scd + (size_t)b * A);   // split images: n_eff for nms_rotated_emit; -1: rows written here
  if (g > 0) {
    const int nn = candidate_total(S.blk_count + (size_t)b * S.nblk, S.nblk, A, reinterpret_cast<int *>(smem + S.scr_off));
    if (nn == 0 || next_pow2(nn) > S.lds_keys) return;   // nothing to do / group 0 takes the whole image (global-memory path)
  }

  int P2 = 0;
  bool in_lds = true;
  int n_total = 0;   // the rotated suppression is not a greedy pass that stops early: it always takes the whole list
  const int n = gather_candidates(S.blk_count + (size_t)b * S.nblk, S.nblk, S.seg, A, S.lds_keys, lkeys, gkeys,
                                  reinterpret_cast<uint64_t *>(S.sbox + (size_t)b * A), reinterpret_cast<int *>(smem + S.scr_off), P2, in_lds,
                                  nullptr, n_total);
  if (n == 0) {
    if (tid == 0) { S.out_count[b] = 0; if (G > 1) *gn = -1; }
    return;
  }
  const bool split = G > 1 && in_lds;
  if (in_lds) bitonic_desc_lds(lkeys, P2);
  else bitonic_desc(gkeys, P2);
  uint64_t *keys = in_lds ? lkeys : gkeys;
  float4 *box = in_lds ? lbox : S.sbox + (size_t)b * A;
  float2 *cd = in_lds ? lcd : S.scd + (size_t)b * A;
  const int n_eff = min(n, S.max_nms);

  for (int i = tid; i < n_eff; i += kSortThreads) {
    const uint32_t anchor = ~(uint32_t)keys[i];
    const float x = ldf(pb, anchor), y = ldf(pb, (size_t)A + anchor), w = ldf(pb, (size_t)2 * A + anchor), h = ldf(pb, (size_t)3 * A + anchor);
    const float ang = ldf(pb, (size_t)(C - 1) * A + anchor);
    const float off = S.agnostic ? 0.0f : (float)S.cls_of[(size_t)b * A + anchor] * S.max_wh;
    const float w2 = w * w / 12.0f, h2 = h * h / 12.0f;
    const float cs = cosf(ang), sn = sinf(ang);
    const float cos2 = cs * cs, sin2 = sn * sn;
    const float ca = w2 * cos2 + h2 * sin2, cb = w2 * sin2 + h2 * cos2, cc = ((w2 - h2) * cs) * sn;
    box[i] = make_float4(x + off, y + off, ca, cb);
    // `reach`: a + b (the trace of the covariance) for a well-conditioned box, +inf otherwise -- see the pair loop
    const float d = fmaxf(ca * cb - cc * cc, 0.0f), tr = ca + cb;
    cd[i] = make_float2(cc, (d >= 1e-3f && d >= 1e-4f * (tr * tr)) ? tr : __builtin_inff());
  }
  __syncthreads();

  // Column j is dropped unless every earlier (higher-scored) box has probiou < thr; the decision is parked in bit 31
  // of the key's anchor word (~anchor has it set: anchors < 2^24), cleared = dropped.
  // `ious.triu_(1).max(0) < thr`: the column maximum includes the zeros below the diagonal, so nothing passes thr <= 0.
  //
  // Far pairs never see the formula (three IEEE divisions, log, exp, two square roots: ~250 instructions).  For two
  // well-conditioned boxes (det >= 1e-4 trace^2 each, so the summed covariance's determinant is far above fp32
  // rounding and above the formula's eps) the distance terms obey t1 + t2 = d'S^-1 d / 4 >= |d|^2 / (4 trace(S)),
  // and t3 >= log(1/2)/2: with |d|^2 > 160 trace(S) the Bhattacharyya distance exceeds 37, exp(-bd) vanishes
  // against the eps under the root and probiou <= 0 < thr -- the full evaluation would say "no hit" too.  Boxes of
  // different classes sit 7680 px apart, so this is nearly every pair; a degenerate box (reach = inf) never takes
  // the shortcut.  The determinant term d is recomputed from a, b, c exactly as it was formed.
  //
  // A wave owns every 16th column and scans its rows 64 at a time with the cheap test; the near pairs go into the
  // wave's queue in LDS and are evaluated 64 at a time, every lane busy with the formula (lane = column, rows walked
  // serially, ran the formula for the whole wave whenever any lane needed it: 4-10 x slower).
  if (!(0.0f < S.iou_thr)) {
    for (int j = tid; j < n_eff; j += kSortThreads) keys[j] &= ~(1ull << 31);
  } else {
    uint32_t *queue = reinterpret_cast<uint32_t *>(smem + S.scr_off + kGatherLds) + wave * kRotQueue;
    uint32_t *key_lo = reinterpret_cast<uint32_t *>(keys);   // little endian: the anchor word of key j is key_lo[2 j]
    int qn = 0;   // queued pairs (wave-uniform)
    auto evaluate = [&](int cnt) {
      if (lane < cnt) {
        const uint32_t pr = queue[lane];
        const int i = (int)(pr & 0xffffu), j = (int)(pr >> 16);
        const float4 bi = box[i], bj = box[j];
        const float2 ci = cd[i], cj = cd[j];
        const float di = fmaxf(bi.z * bi.w - ci.x * ci.x, 0.0f), dj = fmaxf(bj.z * bj.w - cj.x * cj.x, 0.0f);
        const float v = probiou(bi.x, bi.y, bi.z, bi.w, ci.x, di, bj.x, bj.y, bj.z, bj.w, cj.x, dj);
        if (!(v < S.iou_thr)) atomicAnd(key_lo + 2 * j, 0x7fffffffu);
      }
    };
    const int jstep = (kSortThreads / 64) * (split ? G : 1);
    for (int j = wave + (split ? (kSortThreads / 64) * g : 0); j < n_eff; j += jstep) {
      if (!(key_lo[2 * j] >> 31)) continue;   // dropped already by a pair evaluated earlier (a drop is final)
      const float4 bj = box[j];
      const float rj = cd[j].y;
      for (int i0 = 0; i0 < j; i0 += 64) {
        const int i = i0 + lane;
        bool near = false;
        if (i < j) {
          const float4 bi = box[i];
          const float dx = bi.x - bj.x, dy = bi.y - bj.y;
          near = !(dx * dx + dy * dy > 160.0f * (cd[i].y + rj));
        }
        const uint64_t m = __ballot(near);
        if (m == 0) continue;
        if (near) queue[qn + __builtin_popcountll(m & ((1ull << lane) - 1ull))] = (uint32_t)i | ((uint32_t)j << 16);
        qn += __builtin_popcountll(m);
        __builtin_amdgcn_wave_barrier();
        if (qn >= 64) {
          evaluate(64);
          __builtin_amdgcn_wave_barrier();
          const uint32_t t = lane < qn - 64 ? queue[64 + lane] : 0u;
          __builtin_amdgcn_wave_barrier();
          if (lane < qn - 64) queue[lane] = t;
          __builtin_amdgcn_wave_barrier();
          qn -= 64;
        }
      }
    }
    __builtin_amdgcn_wave_barrier();
    evaluate(qn);
  }
  if (split) {
    // the keep flags of this group's columns and (group 0) the sorted keys, for nms_rotated_emit
    __syncthreads();
    unsigned char *gflags = reinterpret_cast<unsigned char *>(S.sbox + (size_t)b * A);   // n_eff bytes, then the keys
    constexpr int NW = kSortThreads / 64;
    for (int j = NW * g + (tid % NW) + NW * G * (tid / NW); j < n_eff; j += NW * G * (kSortThreads / NW))
      gflags[j] = (unsigned char)((keys[j] >> 31) & 1ull);
    if (g == 0) {
      uint64_t *gsorted = reinterpret_cast<uint64_t *>(gflags + ((n_eff + 15) & ~15));
      for (int j = tid; j < n_eff; j += kSortThreads) gsorted[j] = keys[j];
      if (tid == 0) *gn = n_eff;
    }
    return;
  }
  // *gn aliases cd[0] on the global-memory path (in_lds false): every wave must have finished its pair tests, which read
  // cd[0] for the pairs (0, j), before the marker -- a NaN bit pattern -- lands there
  __syncthreads();
  if (tid == 0) { wcnt[16] = 0; if (G > 1) *gn = -1; }
  __syncthreads();

  // ordered compaction of the survivors, first max_det of them
  for (int base = 0; base < n_eff; base += kSortThreads) {
    const int K = wcnt[16];
    if (K >= S.max_det) break;
    const int j = base + tid;
    const bool k = j < n_eff && ((keys[j] >> 31) & 1ull);
    const uint64_t m = __ballot(k);
    if (lane == 0) wcnt[wave] = __builtin_popcountll(m);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < kSortThreads / 64; ++w) {
      const int t = wcnt[w];
      if (w < wave) before += t;
      total += t;
    }
    const int idx = K + before + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (k && idx < S.max_det) kpos[idx] = j;
    __syncthreads();
    if (tid == 0) wcnt[16] = min(K + total, S.max_det);
    __syncthreads();
  }
  const int K = wcnt[16];

  if (tid == 0) S.out_count[b] = K;
  const int row = 6 + S.nm;
  float *od = S.out_det + (size_t)b * S.max_det * row;
  for (int e = tid; e < K * row; e += kSortThreads) {
    const int t = e / row, col = e - t * row;
    const uint64_t key = keys[kpos[t]];
    const uint32_t anchor = ~(uint32_t)key & 0x7fffffffu;
    float v;
    if (col < 4) v = ldf(pb, (size_t)col * A + anchor);
    else if (col == 4) v = sortable2f((uint32_t)(key >> 32));
    else if (col == 5) v = (float)S.cls_of[(size_t)b * A + anchor];
    else v = ldf(pb, (size_t)(4 + S.nc + (col - 6)) * A + anchor);
    od[e] = v;
    if (col == 0) S.out_anchor[(size_t)b * S.max_det + t] = (int)anchor;
  }
}

// Second launch of the split rotated NMS (rot_groups > 1): one workgroup per image compacts the survivors in score
// order from the keep flags the groups left, and writes the rows -- the tail of nms_sort_rotated on the global copies.
template <typename InT>
__global__ __launch_bounds__(kSortThreads) void nms_rotated_emit(SortNmsParams S) {
  __shared__ int kpos[LMI_MAX_DET_CAP];
  __shared__ int wcnt[kSortThreads / 64 + 1];
  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int A = S.A, C = 4 + S.nc + S.nm;
  const int n_eff = *reinterpret_cast<const int *>(S.scd + (size_t)b * A);
  if (n_eff <= 0) return;   // -1: nms_sort_rotated wrote this image's rows itself
  const InT *pb = reinterpret_cast<const InT *>(S.pred) + (size_t)b * C * A;
  const unsigned char *flags = reinterpret_cast<const unsigned char *>(S.sbox + (size_t)b * A);
  const uint64_t *keys = reinterpret_cast<const uint64_t *>(flags + ((n_eff + 15) & ~15));
  constexpr int NW = kSortThreads / 64;
  if (tid == 0) wcnt[NW] = 0;
  __syncthreads();
  for (int base = 0; base < n_eff; base += kSortThreads) {
    const int K = wcnt[NW];
    if (K >= S.max_det) break;
    const int j = base + tid;
    const bool k = j < n_eff && flags[j] != 0;
    const uint64_t m = __ballot(k);
    if (lane == 0) wcnt[wave] = __builtin_popcountll(m);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < NW; ++w) {
      const int t = wcnt[w];
      if (w < wave) before += t;
      total += t;
    }
    const int idx = K + before + __builtin_popcountll(m & ((1ull << lane) - 1ull));
    if (k && idx < S.max_det) kpos[idx] = j;
    __syncthreads();
    if (tid == 0) wcnt[NW] = min(K + total, S.max_det);
    __syncthreads();
  }
  const int K = wcnt[NW];
  if (tid == 0) S.out_count[b] = K;
  const int row = 6 + S.nm;
  float *od = S.out_det + (size_t)b * S.max_det * row;
  for (int e = tid; e < K * row; e += kSortThreads) {
    const int t = e / row, col = e - t * row;
    const uint64_t key = keys[kpos[t]];
    const uint32_t anchor = ~(uint32_t)key & 0x7fffffffu;
    float v;
    if (col < 4) v = ldf(pb, (size_t)col * A + anchor);
    else if (col == 4) v = sortable2f((uint32_t)(key >> 32));
    else if (col == 5) v = (float)S.cls_of[(size_t)b * A + anchor];
    else v = ldf(pb, (size_t)(4 + S.nc + (col - 6)) * A + anchor);
    od[e] = v;
    if (col == 0) S.out_anchor[(size_t)b * S.max_det + t] = (int)anchor;
  }
}

// ---------------------------------------------------------------------------------------------

// grid = batch, block = max_det rounded up to a wave (<= 1024): lane i owns detection i of image b.
__global__ __launch_bounds__(1024) void finalize_boxes_kernel(FinalizeParams F) {
  __shared__ int wtot[16];
  const int b = blockIdx.x;
  const int i = threadIdx.x, lane = i & 63, wave = i >> 6, nwaves = blockDim.x >> 6;
  const int n = min(F.count[b], F.max_det);
  bool k = false;
  if (i < n) {
    const BoxScale bs = box_scale(F, b);
    float *d = F.det + ((size_t)b * F.max_det + i) * F.det_stride;
    const float x1 = d[0], y1 = d[1], x2 = d[2], y2 = d[3];
    const float conf = d[4];
    const int c = (int)d[5];
    d[0] = scale_x(bs, x1);
    d[1] = scale_y(bs, y1);
    d[2] = scale_x(bs, x2);
    d[3] = scale_y(bs, y2);
    const double t = F.thres ? ((c >= 0 && c < F.nc) ? F.thres[c] : 1.0) : F.uniform_thres;
    k = (double)conf > t;
  }
  const uint64_t mask = __ballot(k);
  if (lane == 0) wtot[wave] = __builtin_popcountll(mask);
  __syncthreads();
  int before = 0, total = 0;
  for (int w = 0; w < nwaves; ++w) {
    const int t = wtot[w];
    if (w < wave) before += t;
    total += t;
  }
  if (i < F.max_det) {
    F.keep[(size_t)b * F.max_det + i] = k ? 1 : 0;
    if (F.mask_slot)  // rank inside the image for now; finalize_slots_kernel adds the image's base
      F.mask_slot[(size_t)b * F.max_det + i] = k ? before + __builtin_popcountll(mask & ((1ull << lane) - 1ull)) : -1;
  }
  if (i == 0) F.keep_count[b] = total;
}

// grid = batch: image b sums keep_count[0..b) itself (batch <= 8192 ints, L2-resident), then turns its ranks into
// global slots and fills the slot -> (b,i) map.  One launch of `batch` small blocks instead of one block walking
// batch*max_det entries serially.
__global__ __launch_bounds__(256) void finalize_slots_kernel(FinalizeParams F) {
  __shared__ int wsum[4];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int part = 0;
  for (int j = tid; j < b; j += 256) part += F.keep_count[j];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) part += __shfl_xor(part, d);
  if (lane == 0) wsum[wave] = part;
  __syncthreads();
  const int base = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (tid == 0 && F.slot_base) {
    F.slot_base[b] = base;
    if (b == F.batch - 1) F.slot_base[F.batch] = base + F.keep_count[b];
  }
  if (F.mask_slot) {
    for (int i = tid; i < F.max_det; i += 256) {
      const int e = b * F.max_det + i;
      const int r = F.mask_slot[e];
      if (r >= 0) {
        F.mask_slot[e] = base + r;
        if (F.slot_map) F.slot_map[base + r] = e;
      }
    }
  }
}

// ops.scale_coords on the keypoint columns of the NMS rows, in place: x -= pad_x, y -= pad_y (pads NOT rounded,
// unlike scale_boxes), /= gain, clip to the original frame.  grid = batch, lanes stride over (row, keypoint).
struct CoordsParams {
  float *det;
  const int *count;
  int batch, max_det, det_stride, col0, nkpt, ndim;
  int model_h, model_w;
  const int *orig_hw;
  int uh0, uw0;
};

__global__ __launch_bounds__(256) void scale_coords_kernel(CoordsParams F) {
  const int b = blockIdx.x;
  const int n = min(F.count[b], F.max_det);
  const int h0 = F.orig_hw ? F.orig_hw[2 * b] : F.uh0;
  const int w0 = F.orig_hw ? F.orig_hw[2 * b + 1] : F.uw0;
  const double gh = (double)F.model_h / (double)h0, gw = (double)F.model_w / (double)w0;
  const double gain = gh < gw ? gh : gw;
  const float fpx = (float)(((double)F.model_w - (double)w0 * gain) / 2.0);
  const float fpy = (float)(((double)F.model_h - (double)h0 * gain) / 2.0);
  const float fgain = (float)gain, fw0 = (float)w0, fh0 = (float)h0;
  for (int e = threadIdx.x; e < n * F.nkpt; e += 256) {
    const int i = e / F.nkpt, k = e - i * F.nkpt;
    float *p = F.det + ((size_t)b * F.max_det + i) * F.det_stride + F.col0 + k * F.ndim;
    const float x = (p[0] - fpx) / fgain, y = (p[1] - fpy) / fgain;
    p[0] = fminf(fmaxf(x, 0.0f), fw0);
    p[1] = fminf(fmaxf(y, 0.0f), fh0);
  }
}

// Yolov8Obb.postprocess after NMS (object_detectors/yolov8_lmi/model.py:436-455; formulas in oracle/obb.py):
// regularize_rboxes -> scale_boxes(xywh=True) + clip -> xywhr2xyxyxyxy, and the per-class threshold.  The NMS rows
// are left untouched (the reference builds the corner boxes from a copy).  grid = batch, lane = detection.
struct ObbParams {
  const float *det;
  const int *count;
  int batch, max_det, det_stride, nc, angle_col;
  int model_h, model_w;
  const int *orig_hw;
  int uh0, uw0;
  const double *thres;
  double uniform_thres;
  float *corners;  // (batch, max_det, 4, 2)
  uint8_t *keep;
  int *keep_count;
};

__global__ __launch_bounds__(1024) void finalize_obb_kernel(ObbParams F) {
  __shared__ int wtot[16];
  const int b = blockIdx.x;
  const int i = threadIdx.x, lane = i & 63, wave = i >> 6, nwaves = blockDim.x >> 6;
  const int n = min(F.count[b], F.max_det);
  const int h0 = F.orig_hw ? F.orig_hw[2 * b] : F.uh0;
  const int w0 = F.orig_hw ? F.orig_hw[2 * b + 1] : F.uw0;
  bool k = false;
  if (i < n) {
    const float *d = F.det + ((size_t)b * F.max_det + i) * F.det_stride;
    float x = d[0], y = d[1];
    const float w = d[2], h = d[3], conf = d[4];
    const int c = (int)d[5];
    float t = d[F.angle_col];
    // regularize_rboxes: longer edge first, angle into [0, pi)
    const float kPi = 3.14159265358979323846f, kHalfPi = 1.57079632679489661923f;
    const bool wide = w > h;
    float w_ = wide ? w : h, h_ = wide ? h : w;
    t = wide ? t : t + kHalfPi;
    float m = fmodf(t, kPi);              // torch's % : fmod, then shifted into the divisor's sign
    if (m != 0.0f && m < 0.0f) m += kPi;
    t = m;
    // scale_boxes(xywh=True): pads rounded as for xyxy boxes, only the centre is shifted; clip_boxes clamps all 4 columns
    const double gh = (double)F.model_h / (double)h0, gw = (double)F.model_w / (double)w0;
    const double gain = gh < gw ? gh : gw;
    const float fpx = (float)py_round(((double)F.model_w - (double)w0 * gain) / 2.0 - 0.1);
    const float fpy = (float)py_round(((double)F.model_h - (double)h0 * gain) / 2.0 - 0.1);
    const float fgain = (float)gain, fw0 = (float)w0, fh0 = (float)h0;
    x = fminf(fmaxf((x - fpx) / fgain, 0.0f), fw0);
    y = fminf(fmaxf((y - fpy) / fgain, 0.0f), fh0);
    w_ = fminf(fmaxf(w_ / fgain, 0.0f), fw0);
    h_ = fminf(fmaxf(h_ / fgain, 0.0f), fh0);
    // xywhr2xyxyxyxy
    const float cs = cosf(t), sn = sinf(t);
    const float hw = w_ / 2.0f, nhh = (-h_) / 2.0f, hh = h_ / 2.0f;
    const float v1x = hw * cs, v1y = hw * sn, v2x = nhh * sn, v2y = hh * cs;
    float *o = F.corners + ((size_t)b * F.max_det + i) * 8;
    o[0] = (x + v1x) + v2x; o[1] = (y + v1y) + v2y;
    o[2] = (x + v1x) - v2x; o[3] = (y + v1y) - v2y;
    o[4] = (x - v1x) - v2x; o[5] = (y - v1y) - v2y;
    o[6] = (x - v1x) + v2x; o[7] = (y - v1y) + v2y;
    const double th = F.thres ? ((c >= 0 && c < F.nc) ? F.thres[c] : 1.0) : F.uniform_thres;
    k = (double)conf > th;
  }
  const uint64_t mask = __ballot(k);
  if (lane == 0) wtot[wave] = __builtin_popcountll(mask);
  __syncthreads();
  int total = 0;
  for (int w = 0; w < nwaves; ++w) total += wtot[w];
  if (i < F.max_det) F.keep[(size_t)b * F.max_det + i] = k ? 1 : 0;
  if (i == 0) F.keep_count[b] = total;
}

}  // namespace

// Per-device set-up, done once per GPU on the first NMS call made with that GPU current: the largest dynamic LDS size
// either sort kernel may be launched with (setting the attribute does not change the occupancy of smaller launches)
// and the device address of this GPU's copy of `scratch`.
constexpr int kMaxDevices = 64;
struct NmsDeviceState {
  bool ready = false;
  void *scratch_ptr = nullptr;
  int cu_count = 0;
};
static std::mutex g_nms_dev_mutex;
static NmsDeviceState g_nms_dev[kMaxDevices];

constexpr size_t nms_lds_small(int lds_keys, int max_det) {
  return (size_t)lds_keys * 8 + 4 * 6 * (size_t)((max_det + 1) & ~1) + 64 * 8 + kNmsWindow * 4 + 16;
}
constexpr size_t nms_lds_rot_head(int lds_keys, int max_det) { return (size_t)lds_keys * 8 + 4 * (size_t)((max_det + 1) & ~1) + 4 * 17 + 16; }
constexpr size_t kNmsLdsMax = ((nms_lds_small(kSortLdsCap, LMI_MAX_DET_CAP) + 15) & ~(size_t)15) + (size_t)kSortLdsCap * 16 + kGatherLds;
constexpr size_t kRotQueueLds = (size_t)(kSortThreads / 64) * kRotQueue * 4;
constexpr size_t kNmsLdsRotMax = ((nms_lds_rot_head(kSortLdsCap, LMI_MAX_DET_CAP) + 15) & ~(size_t)15) + (size_t)kSortLdsCap * (16 + 8) + kGatherLds + kRotQueueLds;
static_assert(kNmsLdsMax <= 160 * 1024 && kNmsLdsRotMax <= 160 * 1024, "sort kernels must fit the 160 KiB LDS of a gfx950 CU");

static int nms_device_state(NmsDeviceState **out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    lmi_set_error("lmi_nms: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
    return LMI_ERR_NO_DEVICE;
  }
  std::lock_guard<std::mutex> lock(g_nms_dev_mutex);
  NmsDeviceState &D = g_nms_dev[dev];
  if (!D.ready) {
    e = hipFuncSetAttribute(reinterpret_cast<const void *>(nms_sort_suppress<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNmsLdsMax);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(nms_sort_suppress<__half>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNmsLdsMax);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(nms_sort_rotated<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNmsLdsRotMax);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(nms_sort_rotated<__half>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kNmsLdsRotMax);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(nms_pair_lists<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds_bytes(kPairCapMax));
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void *>(nms_pair_lists<__half>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pair_lds_bytes(kPairCapMax));
    if (e != hipSuccess) { lmi_set_error("lmi_nms: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LMI_ERR_LAUNCH; }
    void *ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || ptr == nullptr) { lmi_set_error("lmi_nms: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e)); return LMI_ERR_LAUNCH; }
    D.scratch_ptr = ptr;
    int cus = 0;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    D.cu_count = (e == hipSuccess && cus > 0) ? cus : 1;
    D.ready = true;
  }
  *out = &D;
  return LMI_OK;
}

extern "C" {

int64_t lmi_nms_workspace_bytes(int batch, int anchors) {
  if (batch <= 0 || anchors <= 0) return 0;
  return (int64_t)nms_layout(batch, anchors).total;
}

struct NmsCall {
  NmsLayout L;
  hipStream_t s;
  int *blk_count;
  uint64_t *keys;
  int *cls_of;
  float4 *sbox;
  unsigned char *ws;
  int half;   // head dtype: 0 fp32, 1 fp16
};

static int nms_check(const void *pred, int pred_dtype, int batch, int nc, int nm, int anchors, void *workspace, int64_t workspace_bytes,
                     void *stream, NmsCall &C) {
  LMI_REQUIRE(pred && workspace, LMI_ERR_BAD_ARG, "lmi_nms_v8: NULL pointer");
  LMI_REQUIRE(pred_dtype == LMI_F32 || pred_dtype == LMI_F16, LMI_ERR_BAD_ARG, "lmi_nms_v8: pred_dtype must be LMI_F32 or LMI_F16");
  LMI_REQUIRE(batch > 0 && batch <= 65535 && nc > 0 && nm >= 0 && anchors > 0, LMI_ERR_BAD_ARG,
              "lmi_nms_v8: bad shape batch=%d nc=%d nm=%d anchors=%d", batch, nc, nm, anchors);
  LMI_REQUIRE(anchors <= (1 << 24), LMI_ERR_LIMIT, "lmi_nms_v8: anchors %d > 2^24", anchors);
  LMI_REQUIRE(lmi_is_aligned(pred, 16) && lmi_is_aligned(workspace, 256), LMI_ERR_ALIGNMENT,
              "lmi_nms_v8: pred must be 16-byte and workspace 256-byte aligned");
  C.L = nms_layout(batch, anchors);
  LMI_REQUIRE(workspace_bytes >= (int64_t)C.L.total, LMI_ERR_WORKSPACE, "lmi_nms_v8: workspace %lld < %lld bytes",
              (long long)workspace_bytes, (long long)C.L.total);
  C.s = reinterpret_cast<hipStream_t>(stream);
  C.ws = reinterpret_cast<unsigned char *>(workspace);
  C.blk_count = reinterpret_cast<int *>(C.ws + C.L.off_count);
  C.keys = reinterpret_cast<uint64_t *>(C.ws + C.L.off_keys);
  C.cls_of = reinterpret_cast<int *>(C.ws + C.L.off_cls);
  C.sbox = reinterpret_cast<float4 *>(C.ws + C.L.off_sbox);
  C.half = pred_dtype == LMI_F16 ? 1 : 0;
  return LMI_OK;
}

static int nms_check_suppress(const float *out_det, const int32_t *out_count, const int32_t *out_anchor, double iou_thres,
                              int max_det, int max_nms) {
  LMI_REQUIRE(out_det && out_count && out_anchor, LMI_ERR_BAD_ARG, "lmi_nms_v8: NULL pointer");
  LMI_REQUIRE(max_det > 0 && max_det <= LMI_MAX_DET_CAP, LMI_ERR_LIMIT, "lmi_nms_v8: max_det %d not in 1..%d", max_det, LMI_MAX_DET_CAP);
  LMI_REQUIRE(max_nms > 0, LMI_ERR_BAD_ARG, "lmi_nms_v8: max_nms must be positive");
  LMI_REQUIRE(iou_thres >= 0.0 && iou_thres <= 1.0, LMI_ERR_BAD_ARG, "lmi_nms_v8: Invalid IoU %f, valid values are between 0.0 and 1.0", iou_thres);
  LMI_REQUIRE(lmi_is_aligned(out_det, 4), LMI_ERR_ALIGNMENT, "lmi_nms_v8: out_det must be 4-byte aligned");
  return LMI_OK;
}

static int fin_check(const lmi_finalize_args *fin, int batch, int rotated, int nm) {
  if (fin == nullptr) return LMI_OK;
  LMI_REQUIRE(fin->kpt_n >= 0 && (fin->kpt_n == 0 || ((fin->kpt_ndim == 2 || fin->kpt_ndim == 3) && fin->kpt_col0 >= 6 &&
                                                       fin->kpt_col0 + fin->kpt_n * fin->kpt_ndim <= 6 + nm)),
              LMI_ERR_BAD_ARG, "lmi_nms_v8_ex: finalize: keypoint columns %d + %d x %d do not fit rows of %d", fin->kpt_col0, fin->kpt_n,
              fin->kpt_ndim, 6 + nm);
  LMI_REQUIRE(!rotated, LMI_ERR_BAD_ARG, "lmi_nms_v8_ex: the fused finalize serves axis-aligned boxes only (use lmi_finalize_obb)");
  LMI_REQUIRE(fin->keep && fin->keep_count, LMI_ERR_BAD_ARG, "lmi_nms_v8_ex: finalize: NULL pointer");
  LMI_REQUIRE(batch <= 8192, LMI_ERR_BAD_ARG, "lmi_nms_v8_ex: finalize: batch %d > 8192", batch);
  LMI_REQUIRE(fin->model_h > 0 && fin->model_w > 0, LMI_ERR_BAD_ARG, "lmi_nms_v8_ex: finalize: bad model size");
  LMI_REQUIRE(fin->orig_hw != nullptr || (fin->uniform_h0 > 0 && fin->uniform_w0 > 0), LMI_ERR_BAD_ARG, "lmi_nms_v8_ex: finalize: original image size missing");
  LMI_REQUIRE(fin->slot_map == nullptr || fin->mask_slot != nullptr, LMI_ERR_BAD_ARG, "lmi_nms_v8_ex: finalize: slot_map needs mask_slot");
  return LMI_OK;
}

// first half: candidate filter (HBM-bound, reads the head once)
static int nms_decode(const NmsCall &C, const void *pred, int batch, int nc, int nm, int anchors, float conf_thres) {
  const float *pf = reinterpret_cast<const float *>(pred);
  const __half *ph = reinterpret_cast<const __half *>(pred);
  if (C.L.seg == 256) {
    dim3 grid(C.L.nblk, batch);
    if (C.half) hipLaunchKernelGGL((nms_decode_filter<4, __half>), grid, dim3(256), 0, C.s, ph, nc, nm, anchors, conf_thres, C.keys, C.L.P, C.cls_of, C.blk_count);
    else hipLaunchKernelGGL((nms_decode_filter<4, float>), grid, dim3(256), 0, C.s, pf, nc, nm, anchors, conf_thres, C.keys, C.L.P, C.cls_of, C.blk_count);
  } else {
    dim3 grid(C.L.nblk, batch);
    if (C.half) hipLaunchKernelGGL((nms_decode_filter<1, __half>), grid, dim3(256), 0, C.s, ph, nc, nm, anchors, conf_thres, C.keys, C.L.P, C.cls_of, C.blk_count);
    else hipLaunchKernelGGL((nms_decode_filter<1, float>), grid, dim3(256), 0, C.s, pf, nc, nm, anchors, conf_thres, C.keys, C.L.P, C.cls_of, C.blk_count);
  }
  return lmi_check_launch("nms_decode_filter");
}

static void fin_params(const lmi_finalize_args *fin, float *det, const int32_t *count, int batch, int max_det, int det_stride, int nc,
                       FinalizeParams &F) {
  F.det = det; F.count = count; F.batch = batch; F.max_det = max_det; F.det_stride = det_stride; F.nc = nc;
  F.model_h = fin->model_h; F.model_w = fin->model_w; F.orig_hw = fin->orig_hw; F.uh0 = fin->uniform_h0; F.uw0 = fin->uniform_w0;
  F.pad_round = fin->pad_round ? 1 : 0; F.thres = fin->thres; F.uniform_thres = fin->uniform_thres; F.keep = fin->keep;
  F.keep_count = fin->keep_count; F.mask_slot = fin->mask_slot; F.slot_map = fin->slot_map; F.slot_base = fin->slot_base;
  F.kpt_col0 = fin->kpt_col0; F.kpt_n = fin->kpt_n; F.kpt_ndim = fin->kpt_ndim;
}

// second half: sort + suppression (+ the fused finalize), one workgroup per image (latency / LDS-bound, next to no HBM traffic)
static int nms_suppress(const NmsCall &C, const void *pred, int batch, int nc, int nm, int anchors, double iou_thres, int agnostic,
                        int max_det, int max_nms, float max_wh, float *out_det, int32_t *out_count, int32_t *out_anchor, int rotated_flags,
                        const lmi_finalize_args *fin) {
  const int rotated = (rotated_flags & 0xff) ? 1 : 0, sweep_only = (rotated_flags & LMI_NMS_SWEEP_ONLY) ? 1 : 0;
  // torchvision compares the fp32 IoU against the threshold as a double: use the largest fp32 <= thr.
  // nms_rotated compares `ious.max(0) < threshold` in fp32 (tensor vs Python float): the threshold rounded to fp32.
  float thr = (float)iou_thres;
  if (!rotated && (double)thr > iou_thres) thr = std::nextafterf(thr, -INFINITY);
  NmsDeviceState *D = nullptr;
  int rc0 = nms_device_state(&D);   // first call on this GPU: LDS limits of the sort kernels
  if (rc0) return rc0;

  SortNmsParams S{};
  S.pred = pred; S.nc = nc; S.nm = nm; S.A = anchors; S.iou_thr = thr; S.agnostic = agnostic ? 1 : 0;
  S.max_det = max_det; S.max_nms = max_nms; S.max_wh = max_wh; S.keys = C.keys; S.P = C.L.P; S.cls_of = C.cls_of;
  S.sbox = C.sbox; S.blk_count = C.blk_count; S.nblk = C.L.nblk; S.seg = C.L.seg; S.out_det = out_det; S.out_count = out_count; S.out_anchor = out_anchor;
  S.lds_keys = C.L.P < kSortLdsCap ? C.L.P : kSortLdsCap;
  S.scd = reinterpret_cast<float2 *>(C.ws + C.L.off_cd);
  S.pair_flags = reinterpret_cast<int *>(C.ws + C.L.off_flag);
  if (rotated) {
    LMI_REQUIRE(nm >= 1, LMI_ERR_BAD_ARG, "lmi_nms_v8_rotated: the angle must be the last of nm >= 1 extra channels");
    // the near-pair queue of nms_sort_rotated packs (i, j) as two 16-bit halves
    LMI_REQUIRE((anchors < max_nms ? anchors : max_nms) <= 65536, LMI_ERR_LIMIT,
                "lmi_nms_v8_rotated: min(anchors %d, max_nms %d) > 65536 candidates per image", anchors, max_nms);
    S.lbox_off = (int)((nms_lds_rot_head(S.lds_keys, max_det) + 15) & ~(size_t)15);
    S.scr_off = (int)((size_t)S.lbox_off + (size_t)S.lds_keys * (16 + 8));
    const size_t lds_rot = (size_t)S.scr_off + kGatherLds + kRotQueueLds;
    // workgroups per image: as many as leave every workgroup of the launch a CU of its own (they are 1024 lanes and
    // ~150 KiB of LDS each), 16 at most; the per-image scratch of the split needs 9 bytes per candidate
    int G = D->cu_count / batch;
    if (G > 16) G = 16;
    if (G < 2 || anchors < 16) G = 1;
    S.rot_groups = G;
    if (C.half) hipLaunchKernelGGL(nms_sort_rotated<__half>, dim3(batch * G), dim3(kSortThreads), lds_rot, C.s, S);
    else hipLaunchKernelGGL(nms_sort_rotated<float>, dim3(batch * G), dim3(kSortThreads), lds_rot, C.s, S);
    int rcr = lmi_check_launch("nms_sort_rotated");
    if (rcr || G == 1) return rcr;
    if (C.half) hipLaunchKernelGGL(nms_rotated_emit<__half>, dim3(batch), dim3(kSortThreads), 0, C.s, S);
    else hipLaunchKernelGGL(nms_rotated_emit<float>, dim3(batch), dim3(kSortThreads), 0, C.s, S);
    return lmi_check_launch("nms_rotated_emit");
  }
  if (fin) {
    S.fin = 1;
    fin_params(fin, out_det, out_count, batch, max_det, 6 + nm, nc, S.F);
  }
  // Suppressor lists first (nms_pair_lists + nms_resolve_lists) where the sorted sweep is slow and CUs would idle: agnostic
  // NMS and models with a handful of classes (classes too large for the class-parallel path), batches that leave every
  // image at least kPairMinGroups workgroups.  The sweep kernel below then only serves the images those two flagged.
  // LMI_NMS_SWEEP_ONLY (measurement, tests): never.
  S.pair_groups = 0;
  if ((agnostic || nc <= 8) && pair_cap(anchors) >= 64 && !sweep_only) {
    int G = D->cu_count / batch;
    if (G > kPairMaxGroups) G = kPairMaxGroups;
    if (G >= kPairMinGroups) {
      S.pair_groups = G;
      const int p2cap = next_pow2(pair_cap(anchors));
      const size_t lds_pairs = pair_lds_bytes(p2cap), lds_res = resolve_lds_bytes(p2cap, max_det, pair_cap(anchors));
      if (C.half) hipLaunchKernelGGL(nms_pair_lists<__half>, dim3(batch * G), dim3(kSortThreads), lds_pairs, C.s, S);
      else hipLaunchKernelGGL(nms_pair_lists<float>, dim3(batch * G), dim3(kSortThreads), lds_pairs, C.s, S);
      int rcp = lmi_check_launch("nms_pair_lists");
      if (rcp) return rcp;
      if (C.half) hipLaunchKernelGGL(nms_resolve_lists<__half>, dim3(batch), dim3(kSortThreads), lds_res, C.s, S);
      else hipLaunchKernelGGL(nms_resolve_lists<float>, dim3(batch), dim3(kSortThreads), lds_res, C.s, S);
      rcp = lmi_check_launch("nms_resolve_lists");
      if (rcp) return rcp;
    }
  }
  S.lbox_off = (int)((nms_lds_small(S.lds_keys, max_det) + 15) & ~(size_t)15);
  S.scr_off = (int)((size_t)S.lbox_off + (size_t)S.lds_keys * 16);
  size_t lds = (size_t)S.scr_off + kGatherLds;
  if (fin && lds < sizeof(int) * 64) lds = sizeof(int) * 64;   // the epilogue's LDS scratch
  if (C.half) hipLaunchKernelGGL(nms_sort_suppress<__half>, dim3(batch), dim3(kSortThreads), lds, C.s, S);
  else hipLaunchKernelGGL(nms_sort_suppress<float>, dim3(batch), dim3(kSortThreads), lds, C.s, S);
  int rc = lmi_check_launch("nms_sort_suppress");
  if (rc || !fin || (fin->mask_slot == nullptr && fin->slot_base == nullptr)) return rc;
  hipLaunchKernelGGL(finalize_slots_kernel, dim3(batch), dim3(256), 0, C.s, S.F);   // ranks -> dense global mask slots
  return lmi_check_launch("finalize_slots_kernel");
}

int lmi_nms_decode_ex(const void *pred, int pred_dtype, int batch, int nc, int nm, int anchors, float conf_thres, void *workspace,
                      int64_t workspace_bytes, void *stream) {
  NmsCall C;
  int rc = nms_check(pred, pred_dtype, batch, nc, nm, anchors, workspace, workspace_bytes, stream, C);
  if (rc) return rc;
  return nms_decode(C, pred, batch, nc, nm, anchors, conf_thres);
}

int lmi_nms_suppress_ex(const void *pred, int pred_dtype, int batch, int nc, int nm, int anchors, double iou_thres, int agnostic,
                        int max_det, int max_nms, float max_wh, int rotated, const lmi_finalize_args *fin, float *out_det,
                        int32_t *out_count, int32_t *out_anchor, void *workspace, int64_t workspace_bytes, void *stream) {
  NmsCall C;
  int rc = nms_check(pred, pred_dtype, batch, nc, nm, anchors, workspace, workspace_bytes, stream, C);
  if (rc) return rc;
  rc = nms_check_suppress(out_det, out_count, out_anchor, iou_thres, max_det, max_nms);
  if (rc) return rc;
  rc = fin_check(fin, batch, rotated & 0xff, nm);
  if (rc) return rc;
  return nms_suppress(C, pred, batch, nc, nm, anchors, iou_thres, agnostic, max_det, max_nms, max_wh, out_det, out_count, out_anchor,
                      rotated, fin);
}

int lmi_nms_v8_ex(const void *pred, int pred_dtype, int batch, int nc, int nm, int anchors, float conf_thres, double iou_thres,
                  int agnostic, int max_det, int max_nms, float max_wh, int rotated, const lmi_finalize_args *fin, float *out_det,
                  int32_t *out_count, int32_t *out_anchor, void *workspace, int64_t workspace_bytes, void *stream) {
  NmsCall C;
  int rc = nms_check(pred, pred_dtype, batch, nc, nm, anchors, workspace, workspace_bytes, stream, C);
  if (rc) return rc;
  rc = nms_check_suppress(out_det, out_count, out_anchor, iou_thres, max_det, max_nms);
  if (rc) return rc;
  rc = fin_check(fin, batch, rotated & 0xff, nm);
  if (rc) return rc;
  rc = nms_decode(C, pred, batch, nc, nm, anchors, conf_thres);
  if (rc) return rc;
  return nms_suppress(C, pred, batch, nc, nm, anchors, iou_thres, agnostic, max_det, max_nms, max_wh, out_det, out_count, out_anchor,
                      rotated, fin);
}

int lmi_nms_decode(const float *pred, int batch, int nc, int nm, int anchors, float conf_thres, void *workspace,
                   int64_t workspace_bytes, void *stream) {
  return lmi_nms_decode_ex(pred, LMI_F32, batch, nc, nm, anchors, conf_thres, workspace, workspace_bytes, stream);
}

int lmi_nms_suppress(const float *pred, int batch, int nc, int nm, int anchors, double iou_thres, int agnostic, int max_det,
                     int max_nms, float max_wh, int rotated, float *out_det, int32_t *out_count, int32_t *out_anchor,
                     void *workspace, int64_t workspace_bytes, void *stream) {
  return lmi_nms_suppress_ex(pred, LMI_F32, batch, nc, nm, anchors, iou_thres, agnostic, max_det, max_nms, max_wh, rotated, nullptr,
                             out_det, out_count, out_anchor, workspace, workspace_bytes, stream);
}

int lmi_nms_v8(const float *pred, int batch, int nc, int nm, int anchors, float conf_thres, double iou_thres,
               int agnostic, int max_det, int max_nms, float max_wh, float *out_det, int32_t *out_count,
               int32_t *out_anchor, void *workspace, int64_t workspace_bytes, void *stream) {
  return lmi_nms_v8_ex(pred, LMI_F32, batch, nc, nm, anchors, conf_thres, iou_thres, agnostic, max_det, max_nms, max_wh, 0, nullptr,
                       out_det, out_count, out_anchor, workspace, workspace_bytes, stream);
}

int lmi_nms_v8_static(const float *pred, int batch, int nc, int nm, int anchors, float conf_thres, double iou_thres,
                      int agnostic, int max_det, int max_nms, float max_wh, float *out_det, int32_t *out_count,
                      int32_t *out_anchor, void *stream) {
  LMI_REQUIRE(batch > 0 && anchors > 0, LMI_ERR_BAD_ARG, "lmi_nms_v8_static: bad shape batch=%d anchors=%d", batch, anchors);
  const int64_t need = lmi_nms_workspace_bytes(batch, anchors);
  LMI_REQUIRE(need <= (int64_t)LMI_NMS_STATIC_BYTES, LMI_ERR_LIMIT,
              "lmi_nms_v8_static: batch %d x %d anchors needs %lld bytes of workspace, the built-in one holds %lld "
              "(use lmi_nms_v8 with a caller-allocated workspace)", batch, anchors, (long long)need, (long long)LMI_NMS_STATIC_BYTES);
  NmsDeviceState *D = nullptr;
  int rc = nms_device_state(&D);
  if (rc) return rc;
  // same kernels as lmi_nms_v8, on this GPU's copy of `scratch`; the decode half clears the candidate counts, every
  // other part of the workspace is written before it is read
  return lmi_nms_v8_ex(pred, LMI_F32, batch, nc, nm, anchors, conf_thres, iou_thres, agnostic, max_det, max_nms, max_wh, 0, nullptr,
                       out_det, out_count, out_anchor, D->scratch_ptr, (int64_t)LMI_NMS_STATIC_BYTES, stream);
}

int lmi_nms_v8_rotated(const float *pred, int batch, int nc, int nm, int anchors, float conf_thres, double iou_thres,
                       int agnostic, int max_det, int max_nms, float max_wh, float *out_det, int32_t *out_count,
                       int32_t *out_anchor, void *workspace, int64_t workspace_bytes, void *stream) {
  return lmi_nms_v8_ex(pred, LMI_F32, batch, nc, nm, anchors, conf_thres, iou_thres, agnostic, max_det, max_nms, max_wh, 1, nullptr,
                       out_det, out_count, out_anchor, workspace, workspace_bytes, stream);
}

int lmi_finalize_detections(float *det, const int32_t *count, int batch, int max_det, int det_stride, int nc, int model_h,
                            int model_w, const int32_t *orig_hw, int uniform_h0, int uniform_w0, int pad_round,
                            const double *thres, double uniform_thres, uint8_t *keep, int32_t *keep_count,
                            int32_t *mask_slot, int32_t *slot_map, int32_t *slot_base, void *stream) {
  LMI_REQUIRE(det && count && keep && keep_count, LMI_ERR_BAD_ARG, "lmi_finalize_detections: NULL pointer");
  LMI_REQUIRE(batch > 0 && batch <= 8192 && max_det > 0 && det_stride >= 6 && nc > 0, LMI_ERR_BAD_ARG,
              "lmi_finalize_detections: bad shape batch=%d max_det=%d det_stride=%d nc=%d", batch, max_det, det_stride, nc);
  LMI_REQUIRE(model_h > 0 && model_w > 0, LMI_ERR_BAD_ARG, "lmi_finalize_detections: bad model size");
  LMI_REQUIRE(orig_hw != nullptr || (uniform_h0 > 0 && uniform_w0 > 0), LMI_ERR_BAD_ARG, "lmi_finalize_detections: original image size missing");
  LMI_REQUIRE(slot_map == nullptr || mask_slot != nullptr, LMI_ERR_BAD_ARG, "lmi_finalize_detections: slot_map needs mask_slot");
  FinalizeParams F;
  F.det = det; F.count = count; F.batch = batch; F.max_det = max_det; F.det_stride = det_stride; F.nc = nc;
  F.model_h = model_h; F.model_w = model_w; F.orig_hw = orig_hw; F.uh0 = uniform_h0; F.uw0 = uniform_w0;
  F.pad_round = pad_round ? 1 : 0; F.thres = thres; F.uniform_thres = uniform_thres; F.keep = keep; F.keep_count = keep_count;
  F.mask_slot = mask_slot; F.slot_map = slot_map; F.slot_base = slot_base;
  F.kpt_col0 = 0; F.kpt_n = 0; F.kpt_ndim = 0;
  LMI_REQUIRE(max_det <= LMI_MAX_DET_CAP, LMI_ERR_LIMIT, "lmi_finalize_detections: max_det %d > %d", max_det, LMI_MAX_DET_CAP);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(finalize_boxes_kernel, dim3(batch), dim3((max_det + 63) / 64 * 64), 0, s, F);
  int rc = lmi_check_launch("finalize_boxes_kernel");
  if (rc) return rc;
  if (mask_slot == nullptr && slot_base == nullptr) return LMI_OK;
  hipLaunchKernelGGL(finalize_slots_kernel, dim3(batch), dim3(256), 0, s, F);
  return lmi_check_launch("finalize_slots_kernel");
}

int lmi_scale_coords(float *det, const int32_t *count, int batch, int max_det, int det_stride, int col0, int nkpt, int ndim,
                     int model_h, int model_w, const int32_t *orig_hw, int uniform_h0, int uniform_w0, void *stream) {
  LMI_REQUIRE(det && count, LMI_ERR_BAD_ARG, "lmi_scale_coords: NULL pointer");
  LMI_REQUIRE(batch > 0 && batch <= 65535 && max_det > 0 && nkpt > 0 && (ndim == 2 || ndim == 3) && col0 >= 0 &&
                  det_stride >= col0 + nkpt * ndim,
              LMI_ERR_BAD_ARG, "lmi_scale_coords: bad shape batch=%d max_det=%d det_stride=%d col0=%d kpt=%dx%d", batch, max_det,
              det_stride, col0, nkpt, ndim);
  LMI_REQUIRE(model_h > 0 && model_w > 0, LMI_ERR_BAD_ARG, "lmi_scale_coords: bad model size");
  LMI_REQUIRE(orig_hw != nullptr || (uniform_h0 > 0 && uniform_w0 > 0), LMI_ERR_BAD_ARG, "lmi_scale_coords: original image size missing");
  CoordsParams F;
  F.det = det; F.count = count; F.batch = batch; F.max_det = max_det; F.det_stride = det_stride; F.col0 = col0;
  F.nkpt = nkpt; F.ndim = ndim; F.model_h = model_h; F.model_w = model_w; F.orig_hw = orig_hw; F.uh0 = uniform_h0; F.uw0 = uniform_w0;
  hipLaunchKernelGGL(scale_coords_kernel, dim3(batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), F);
  return lmi_check_launch("scale_coords_kernel");
}

int lmi_finalize_obb(const float *det, const int32_t *count, int batch, int max_det, int det_stride, int nc, int model_h,
                     int model_w, const int32_t *orig_hw, int uniform_h0, int uniform_w0, const double *thres,
                     double uniform_thres, float *corners, uint8_t *keep, int32_t *keep_count, void *stream) {
  LMI_REQUIRE(det && count && corners && keep && keep_count, LMI_ERR_BAD_ARG, "lmi_finalize_obb: NULL pointer");
  LMI_REQUIRE(batch > 0 && batch <= 65535 && max_det > 0 && det_stride >= 7 && nc > 0, LMI_ERR_BAD_ARG,
              "lmi_finalize_obb: bad shape batch=%d max_det=%d det_stride=%d nc=%d", batch, max_det, det_stride, nc);
  LMI_REQUIRE(max_det <= LMI_MAX_DET_CAP, LMI_ERR_LIMIT, "lmi_finalize_obb: max_det %d > %d", max_det, LMI_MAX_DET_CAP);
  LMI_REQUIRE(model_h > 0 && model_w > 0, LMI_ERR_BAD_ARG, "lmi_finalize_obb: bad model size");
  LMI_REQUIRE(orig_hw != nullptr || (uniform_h0 > 0 && uniform_w0 > 0), LMI_ERR_BAD_ARG, "lmi_finalize_obb: original image size missing");
  ObbParams F;
  F.det = det; F.count = count; F.batch = batch; F.max_det = max_det; F.det_stride = det_stride; F.nc = nc;
  F.angle_col = det_stride - 1;  // the angle is the last extra channel of the row
  F.model_h = model_h; F.model_w = model_w; F.orig_hw = orig_hw; F.uh0 = uniform_h0; F.uw0 = uniform_w0;
  F.thres = thres; F.uniform_thres = uniform_thres; F.corners = corners; F.keep = keep; F.keep_count = keep_count;
  hipLaunchKernelGGL(finalize_obb_kernel, dim3(batch), dim3((max_det + 63) / 64 * 64), 0, reinterpret_cast<hipStream_t>(stream), F);
  return lmi_check_launch("finalize_obb_kernel");
}

}  // extern "C"
