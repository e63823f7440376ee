/* A plain C99 client of include/lmi_hip.h: the boundary is a C ABI (no C++ types, no torch types).  Built and run by
 * tests/test_abi_host.py with gcc; calls host-only entry points, so it needs no GPU. */
#include <stdio.h>
#include <string.h>

#include "lmi_hip.h"

int main(void) {
  int32_t g[6];
  lmi_frame f;
  lmi_mask_image m;
  lmi_finalize_args a;
  memset(&f, 0, sizeof f); memset(&m, 0, sizeof m); memset(&a, 0, sizeof a);
  if (lmi_version() <= 0) return 1;
  if (lmi_letterbox_geometry(1080, 1920, 640, 640, 0, g) != LMI_OK) return 2;
  printf("%d %d %d %d %d %d\n", g[0], g[1], g[2], g[3], g[4], g[5]);
  printf("%lld %lld %lld\n", (long long)lmi_nms_workspace_bytes(64, 8400), (long long)lmi_mask_workspace_bytes(300, 160, 160),
         (long long)lmi_letterbox_workspace_bytes(1, 640, 640));
  /* argument validation happens before any GPU call: NULL pointers are refused with a message */
  if (lmi_nms_v8(NULL, 1, 80, 0, 8400, 0.25f, 0.45, 0, 300, 30000, 7680.0f, NULL, NULL, NULL, NULL, 0, NULL) != LMI_ERR_BAD_ARG) return 3;
  if (strlen(lmi_last_error()) == 0) return 4;
  if (lmi_letterbox_u8(&f, &f, 1, NULL, 640, 640, LMI_F32, 114, 0, LMI_RESIZE_CV2_LINEAR | LMI_SRC_GRAY, NULL, 0, NULL) != LMI_ERR_BAD_ARG) return 5;
  printf("%s\n", lmi_status_string(LMI_ERR_BAD_ARG));
  return 0;
}
