// Status / error plumbing and host-only helpers of liblmi_hip.so.
#include <cmath>
#include <cstdarg>
#include <cstdio>

#include "common.h"

static thread_local char g_err[512] = "";

void lmi_set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int lmi_check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    lmi_set_error("%s: %s", what, hipGetErrorString(e));
    return LMI_ERR_LAUNCH;
  }
  return LMI_OK;
}

extern "C" {

int lmi_version(void) { return LMI_VERSION; }

const char *lmi_last_error(void) { return g_err; }

const char *lmi_status_string(int s) {
  switch (s) {
    case LMI_OK: return "ok";
    case LMI_ERR_BAD_ARG: return "bad argument";
    case LMI_ERR_ALIGNMENT: return "alignment requirement not met";
    case LMI_ERR_WORKSPACE: return "workspace too small";
    case LMI_ERR_LIMIT: return "size beyond a compiled-in limit";
    case LMI_ERR_LAUNCH: return "HIP launch/runtime error";
    case LMI_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
  }
}

int lmi_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    lmi_set_error("hipGetDeviceCount: %s", hipGetErrorString(e));
    return LMI_ERR_NO_DEVICE;
  }
  return n;
}

// Python's round(): to nearest, ties to even (the default FP environment's nearbyint).
static inline long py_round(double x) { return (long)std::nearbyint(x); }

int lmi_letterbox_geometry(int h, int w, int dst_h, int dst_w, int stretch, int32_t out6[6]) {
  LMI_REQUIRE(out6 != nullptr, LMI_ERR_BAD_ARG, "lmi_letterbox_geometry: out6 is NULL");
  LMI_REQUIRE(h > 0 && w > 0 && dst_h > 0 && dst_w > 0, LMI_ERR_BAD_ARG,
              "lmi_letterbox_geometry: non-positive size (%d,%d)->(%d,%d)", h, w, dst_h, dst_w);
  if (stretch) {
    out6[0] = dst_w; out6[1] = dst_h; out6[2] = out6[3] = out6[4] = out6[5] = 0;
    return LMI_OK;
  }
  double rh = (double)dst_h / (double)h, rw = (double)dst_w / (double)w;
  double r = rh < rw ? rh : rw;
  long nw = py_round((double)w * r), nh = py_round((double)h * r);
  if (nw < 1) nw = 1;
  if (nh < 1) nh = 1;
  if (nw > dst_w) nw = dst_w;
  if (nh > dst_h) nh = dst_h;
  double dw = (double)(dst_w - nw) / 2.0, dh = (double)(dst_h - nh) / 2.0;
  out6[0] = (int32_t)nw;
  out6[1] = (int32_t)nh;
  out6[2] = (int32_t)py_round(dw - 0.1);
  out6[3] = (int32_t)py_round(dw + 0.1);
  out6[4] = (int32_t)py_round(dh - 0.1);
  out6[5] = (int32_t)py_round(dh + 0.1);
  return LMI_OK;
}

}  // extern "C"
