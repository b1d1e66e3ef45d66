// capi_common.h — error reporting shared by the translation units of libnphm_amd.so
#pragma once
#include <hip/hip_runtime.h>
#include <stdio.h>

#include "../../include/nphm_amd.h"

inline char* nphm_err_buf() {
  static thread_local char buf[512] = "";
  return buf;
}
inline int nphm_fail(const char* what, hipError_t e) {
  snprintf(nphm_err_buf(), 512, "%s: %s", what, hipGetErrorString(e));
  return -1;
}
inline int nphm_fail_msg(const char* what) {
  snprintf(nphm_err_buf(), 512, "%s", what);
  return -2;
}
