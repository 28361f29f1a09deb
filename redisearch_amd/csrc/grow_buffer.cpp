// grow_buffer.cpp -- see grow_buffer.hpp.
#include "grow_buffer.hpp"

#include <algorithm>
#include <mutex>

namespace rsgpu {

bool vmm_supported(int device) {
  static std::mutex mu;
  static int cache[64];  // 0 unknown, 1 yes, 2 no
  std::lock_guard<std::mutex> g(mu);
  if (device < 0 || device >= 64) return false;
  if (!cache[device]) {
    int v = 0;
    hipError_t e = hipDeviceGetAttribute(&v, hipDeviceAttributeVirtualMemoryManagementSupported, device);
    cache[device] = (e == hipSuccess && v) ? 1 : 2;
    if (e != hipSuccess) (void)hipGetLastError();
  }
  return cache[device] == 1;
}

static hipMemAllocationProp chunk_prop(int device) {
  hipMemAllocationProp p{};
  p.type = hipMemAllocationTypePinned;
  p.location.type = hipMemLocationTypeDevice;
  p.location.id = device;
  return p;
}

void GrowBuffer::release() {
  if (!ptr_) return;
  if (mapped_) {
    for (const Chunk &c : chunks_) {
      (void)hipMemUnmap(ptr_ + c.offset, c.size);
      (void)hipMemRelease(c.handle);
    }
    (void)hipMemAddressFree(ptr_, va_size_);
    chunks_.clear();
  } else {
    (void)hipFree(ptr_);
  }
  ptr_ = nullptr;
  cap_ = 0;
  va_size_ = 0;
  mapped_ = false;
}

// A (larger) virtual range; the chunks mapped so far move into it -- same physical memory, new addresses.
void GrowBuffer::reserve_va(size_t need) {
  const size_t want = round_up(std::max<size_t>(need * 4, 4ull << 30), chunk_);
  void *np = nullptr;
  HIP_CHECK(hipMemAddressReserve(&np, want, 0, nullptr, 0));
  uint8_t *nptr = static_cast<uint8_t *>(np);
  if (!chunks_.empty()) {
    for (const Chunk &c : chunks_) HIP_CHECK(hipMemUnmap(ptr_ + c.offset, c.size));
    for (const Chunk &c : chunks_) HIP_CHECK(hipMemMap(nptr + c.offset, c.size, 0, c.handle, 0));
    hipMemAccessDesc acc{};
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device_;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    HIP_CHECK(hipMemSetAccess(nptr, cap_, &acc, 1));
  }
  if (ptr_ && va_size_) HIP_CHECK(hipMemAddressFree(ptr_, va_size_));
  ptr_ = nptr;
  va_size_ = want;
}

void GrowBuffer::map_more(size_t bytes) {
  const hipMemAllocationProp prop = chunk_prop(device_);
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device_;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  while (cap_ < bytes) {
    // one physical allocation per step, at most kMaxChunk: this driver maps 1 and 2 GiB handles fine, a 3.75 GiB one
    // faults on access (scripts/diag/vmm_probe.cpp, ROCm 7.0.2); if even that is not available in one piece,
    // minimum-size pieces
    size_t want = std::min<size_t>(round_up(bytes - cap_, chunk_), round_up(kMaxChunk, chunk_));
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, want, &prop, 0);
    if (e != hipSuccess && want > chunk_) {
      (void)hipGetLastError();
      want = chunk_;
      e = hipMemCreate(&h, want, &prop, 0);
    }
    if (e != hipSuccess) throw HipError(e, "hipMemCreate", __FILE__, __LINE__);  // HBM exhausted: what is mapped stays valid
    uint8_t *at = ptr_ + cap_;
    e = hipMemMap(at, want, 0, h, 0);
    if (e == hipSuccess) e = hipMemSetAccess(at, want, &acc, 1);
    if (e != hipSuccess) {
      (void)hipMemRelease(h);
      throw HipError(e, "hipMemMap/hipMemSetAccess", __FILE__, __LINE__);
    }
    chunks_.push_back(Chunk{h, cap_, want});
    cap_ += want;
  }
}

void GrowBuffer::ensure(int device, size_t bytes, size_t live_bytes, hipStream_t s, int mode) {
  if (bytes <= cap_) return;
  device_ = device;
  const bool go_mapped = mapped_ || (mode == 1 && bytes >= kVmmThreshold && vmm_supported(device));
  if (!go_mapped) {  // small (or VMM switched off): a fresh allocation and one copy
    uint8_t *n = nullptr;
    HIP_CHECK(hipMalloc((void **)&n, bytes));
    if (ptr_ && live_bytes) {
      hipError_t e = hipMemcpyAsync(n, ptr_, live_bytes, hipMemcpyDeviceToDevice, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        (void)hipFree(n);
        throw HipError(e, "grow copy", __FILE__, __LINE__);
      }
    }
    if (ptr_) HIP_CHECK(hipFree(ptr_));
    ptr_ = n;
    cap_ = bytes;
    return;
  }
  if (!mapped_) {  // the one migration of this buffer's life
    uint8_t *old = ptr_;
    const size_t old_cap = cap_;
    ptr_ = nullptr;
    cap_ = 0;
    mapped_ = true;
    try {
      size_t gran = 0;
      const hipMemAllocationProp prop = chunk_prop(device);
      HIP_CHECK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
      chunk_ = round_up(kChunk, gran ? gran : (2u << 20));
      reserve_va(bytes);
      map_more(bytes);
      if (old && live_bytes) {
        HIP_CHECK(hipMemcpyAsync(ptr_, old, live_bytes, hipMemcpyDeviceToDevice, s));
        HIP_CHECK(hipStreamSynchronize(s));
      }
    } catch (...) {  // back to the old allocation, which is still intact
      release();
      ptr_ = old;
      cap_ = old_cap;
      mapped_ = false;
      throw;
    }
    if (old) HIP_CHECK(hipFree(old));
    return;
  }
  if (bytes > va_size_) reserve_va(bytes);
  map_more(bytes);
}

}  // namespace rsgpu
