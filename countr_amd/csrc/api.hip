// Library management + error plumbing of libcountr_hip.so (see include/countr_hip.h).
#include "common.hpp"
#include "../../include/countr_hip.h"
#include <string.h>
#include <stdio.h>
#include <atomic>
#include <mutex>

namespace {
thread_local char g_err[512] = "";
// Per-device read-only data, allocated ONCE by countr_init(device) (the only place this library allocates) and never written again:
// a vector of zeros that bias-less launches (input gradients) hand to epilogues that always add a bias.  Launch paths only read the
// pointer (countr_zero_vec): no allocation, no first-touch race between the forward thread and the autograd thread, legal under
// stream capture, one vector per device.
constexpr int MAX_DEV = 64;
std::mutex g_init_mu;
std::atomic<float*> g_zero[MAX_DEV];
}

// Every nullptr return leaves its own message in countr_last_error(): a caller that turns the nullptr into a negative return code must
// never surface a stale text of an earlier, unrelated failure.
const float* countr_zero_vec(int n) {
  int d = -1;
  if (n > COUNTR_ZERO_VEC_FLOATS) {
    char buf[160];
    snprintf(buf, sizeof(buf), "bias-less launch with N = %d: the per-device vector of zeros holds %d floats", n, (int)COUNTR_ZERO_VEC_FLOATS);
    countr_set_error(buf);
    return nullptr;
  }
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= MAX_DEV) {
    (void)hipGetLastError();
    countr_set_error("countr_zero_vec: hipGetDevice failed or the current device index is outside this library's table (64 devices)");
    return nullptr;
  }
  const float* z = g_zero[d].load(std::memory_order_acquire);
  if (!z) countr_set_error("countr_init(device) has not been called for the current device IN THIS LIBRARY (it allocates the per-device constants; "
                           "libcountr_hip.so and libcountr_hip_f16.so each keep their own)");
  return z;
}

extern "C" void countr_set_error(const char* msg) {
  strncpy(g_err, msg ? msg : "", sizeof(g_err) - 1);
  g_err[sizeof(g_err) - 1] = 0;
}

int countr_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    char buf[512];
    snprintf(buf, sizeof(buf), "%s: launch failed: %s", what, hipGetErrorString(e));
    countr_set_error(buf);
    return -10;
  }
  return 0;
}

extern "C" const char* countr_last_error(void) { return g_err; }
thread_local int countr_dry_run = 0;

extern "C" int countr_version(void) { return 9; }   // 2: countr_gemm_args grew (ln_* fields, rowsum_slabs); 3: prefetch hint; 4: countr_gemm_group; 5: countr_step_prologue; 6: countr_softmax_fwd_ld; 7: *_amp (device-side GradScaler); 8: countr_transpose16; 9: countr_gemm_args.gn_rows + countr_groupnorm_relu_fwd_rows

extern "C" int countr_init(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    countr_set_error("countr_init: no HIP device visible (the HIP path has no CPU fallback)");
    return -1;
  }
  if (device < 0 || device >= n) { countr_set_error("countr_init: device index out of range"); return -1; }
  if (hipSetDevice(device) != hipSuccess) { countr_set_error("countr_init: hipSetDevice failed"); return -1; }
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, device) != hipSuccess) { countr_set_error("countr_init: cannot query device"); return -1; }
  if (strncmp(p.gcnArchName, "gfx950", 6) != 0) {
    char buf[256];
    snprintf(buf, sizeof(buf), "countr_init: device arch %s is not gfx950 (kernels are built for MI355X only)", p.gcnArchName);
    countr_set_error(buf);
    return -2;
  }
  if (device >= MAX_DEV) { countr_set_error("countr_init: more than 64 devices"); return -1; }
  std::lock_guard<std::mutex> lk(g_init_mu);
  if (!g_zero[device].load(std::memory_order_acquire)) {
    float* z = nullptr;
    if (hipMalloc(&z, COUNTR_ZERO_VEC_FLOATS * sizeof(float)) != hipSuccess || hipMemset(z, 0, COUNTR_ZERO_VEC_FLOATS * sizeof(float)) != hipSuccess ||
        hipDeviceSynchronize() != hipSuccess) {
      countr_set_error("countr_init: cannot allocate the per-device constants");
      return -3;
    }
    g_zero[device].store(z, std::memory_order_release);
  }
  return 0;
}
