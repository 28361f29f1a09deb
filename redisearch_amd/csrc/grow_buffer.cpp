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

// The virtual range of a mapped buffer: kReserveFactor x what is needed now, at least 64 GiB, at most the device's
// memory -- generous because virtual address space is cheap (47 bits) and because outgrowing it costs a copy:
// re-mapping the same physical chunks into a larger range was tried and is NOT reliable on this driver (ROCm 7.0.2:
// after the move, a new chunk whose end lies 4 GiB or more above the new base is refused with "invalid argument",
// scripts/diag/vmm_probe.cpp and profiles/r02_vmm_notes.txt), so a buffer that does outgrow its range is rebuilt in a
// fresh one (GrowBuffer::ensure).
void GrowBuffer::reserve_va(size_t need) {
  size_t total = 0, free_b = 0;
  (void)hipMemGetInfo(&free_b, &total);
  size_t want = std::max<size_t>(need * (size_t)reserve_factor, (size_t)reserve_factor << 30);  // factor 64: >= 64 GiB
  const size_t device_cap = total + ((size_t)4 << 30);
  if (total && want > device_cap) want = std::max<size_t>(need, device_cap);
  want = round_up(want, chunk_);
  void *np = nullptr;
  HIP_CHECK(hipMemAddressReserve(&np, want, 0, nullptr, 0));
  ptr_ = static_cast<uint8_t *>(np);
  va_size_ = want;
}

void GrowBuffer::map_more(size_t bytes) {
  const hipMemAllocationProp prop = chunk_prop(device_);
  hipMemAccessDesc acc{};
  acc.location.type = hipMemLocationTypeDevice;
  acc.location.id = device_;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  // UNIFORM chunks, each at a multiple of the chunk size from the base: the only shape this driver (ROCm 7.0.2) maps
  // reliably -- with mixed sizes hipMemSetAccess refuses some chunks beyond +4 GiB with "invalid argument"
  // (scripts/diag/vmm_probe3.cpp; profiles/r02_vmm_notes.txt)
  while (cap_ < bytes) {
    hipMemGenericAllocationHandle_t h;
    HIP_CHECK(hipMemCreate(&h, chunk_, &prop, 0));  // HBM exhausted: throws, what is mapped stays valid
    uint8_t *at = ptr_ + cap_;
    hipError_t e = hipMemMap(at, chunk_, 0, h, 0);
    if (e == hipSuccess) {
      e = hipMemSetAccess(at, chunk_, &acc, 1);
      if (e != hipSuccess) (void)hipMemUnmap(at, chunk_);
    }
    if (e != hipSuccess) {
      (void)hipMemRelease(h);
      (void)hipGetLastError();
      char what[256];
      snprintf(what, sizeof what, "hipMemMap/hipMemSetAccess(base %p + %zu, size %zu; reservation %zu, %zu chunks)",
               (void *)ptr_, cap_, chunk_, va_size_, chunks_.size());
      throw HipError(e, what, __FILE__, __LINE__);
    }
    chunks_.push_back(Chunk{h, cap_, chunk_});
    cap_ += chunk_;
  }
}

void GrowBuffer::ensure(int device, size_t bytes, size_t live_bytes, hipStream_t s, int mode) {
  if (bytes <= cap_) return;
  device_ = device;
  const bool go_mapped = mapped_ || (mode == 1 && !vmm_broken && bytes >= kVmmThreshold && vmm_supported(device));
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
      // one chunk size for the life of the buffer: 1 GiB for buffers that are already large when they are mapped
      // (a Reserve()d corpus), 256 MiB for those that grow from small
      const size_t base_chunk = chunk_override ? chunk_override : (bytes >= ((size_t)8 << 30) ? kMaxChunk : kChunk);
      chunk_ = round_up(base_chunk, gran ? gran : (2u << 20));
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
  if (bytes > va_size_) {
    // outgrew the reservation (more than reserve_factor x the size it had when it was mapped): rebuild in a fresh,
    // larger range -- the one case after the migration that copies
    GrowBuffer fresh;
    fresh.reserve_factor = reserve_factor;
    fresh.device_ = device;
    fresh.chunk_ = chunk_;
    fresh.mapped_ = true;
    fresh.reserve_va(bytes);
    try {
      fresh.map_more(bytes);
    } catch (const HipError &e) {
      char what[512];
      snprintf(what, sizeof what, "%s [rebuilding: old base %p range %zu mapped %zu]", e.what(), (void *)ptr_, va_size_, cap_);
      throw HipError(e.code, what, __FILE__, __LINE__);
    }
    if (live_bytes) {
      HIP_CHECK(hipMemcpyAsync(fresh.ptr_, ptr_, live_bytes, hipMemcpyDeviceToDevice, s));
      HIP_CHECK(hipStreamSynchronize(s));
    }
    release();
    ptr_ = fresh.ptr_;
    cap_ = fresh.cap_;
    va_size_ = fresh.va_size_;
    chunks_.swap(fresh.chunks_);
    mapped_ = true;
    fresh.ptr_ = nullptr;  // ownership moved
    fresh.cap_ = 0;
    fresh.mapped_ = false;
    return;
  }
  try {
    map_more(bytes);
  } catch (const HipError &e) {
    if (e.code == hipErrorOutOfMemory) throw;
    // the driver refused a mapping it should have accepted: keep the data, leave the mapped world for good
    uint8_t *n = nullptr;
    HIP_CHECK(hipMalloc((void **)&n, bytes));
    if (live_bytes) {
      hipError_t c = hipMemcpyAsync(n, ptr_, live_bytes, hipMemcpyDeviceToDevice, s);
      if (c == hipSuccess) c = hipStreamSynchronize(s);
      if (c != hipSuccess) {
        (void)hipFree(n);
        throw HipError(c, "grow copy (leaving mapped mode)", __FILE__, __LINE__);
      }
    }
    release();
    ptr_ = n;
    cap_ = bytes;
    mapped_ = false;
    vmm_broken = true;
  }
}

}  // namespace rsgpu
