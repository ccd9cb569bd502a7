// Per-device launch-time facts shared by the launchers of libcmf_amd.so.
//
// hipFuncSetAttribute(MaxDynamicSharedMemorySize) and the CU count are properties of a (device, kernel) pair, not of the
// process: a launcher that remembers "already set" in one static flag leaves the kernel at the 64 KB default on the second
// device of the same process (one Python thread per GPU is the threading contract of SURVEY 8b) and its first launch there
// fails.  The memo below is keyed by the CALLING THREAD'S CURRENT DEVICE.  It is the library's only mutable global: entries
// are published with one atomic compare-exchange, a lost race repeats an idempotent hipFuncSetAttribute, nothing is ever
// removed, so calls stay re-entrant.
#include <atomic>

#include "common.h"

namespace {

constexpr int MAX_DEV = 64, MAX_FN = 128;

struct Slot {
  std::atomic<const void*> fn{nullptr};
  std::atomic<int> bytes{0};
};

Slot g_lds[MAX_DEV][MAX_FN];
std::atomic<int> g_cus[MAX_DEV];

}  // namespace

hipError_t cmf_set_dynamic_lds(const void* fn, int bytes) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= MAX_DEV) return hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  Slot* row = g_lds[dev];
  Slot* mine = nullptr;
  for (int i = 0; i < MAX_FN; ++i) {
    const void* f = row[i].fn.load(std::memory_order_acquire);
    if (f == fn) {
      mine = &row[i];
      break;
    }
    if (f == nullptr) {
      const void* expect = nullptr;
      if (row[i].fn.compare_exchange_strong(expect, fn, std::memory_order_acq_rel) || expect == fn) {
        mine = &row[i];
        break;
      }
    }
  }
  if (mine && mine->bytes.load(std::memory_order_acquire) >= bytes) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess && mine) {
    int cur = mine->bytes.load(std::memory_order_relaxed);
    while (cur < bytes && !mine->bytes.compare_exchange_weak(cur, bytes, std::memory_order_acq_rel)) {
    }
  }
  return e;
}

int cmf_device_cus() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 256;
  if (dev >= 0 && dev < MAX_DEV) {
    const int c = g_cus[dev].load(std::memory_order_acquire);
    if (c > 0) return c;
  }
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  if (dev >= 0 && dev < MAX_DEV) g_cus[dev].store(cus, std::memory_order_release);
  return cus;
}
