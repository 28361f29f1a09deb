// diagnostic 2: many 1 GiB chunks behind one reservation -- where does mapping stop working, and does the
// reservation's alignment or the way access is granted matter?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
static const char *E(hipError_t e) { if (e != hipSuccess) (void)hipGetLastError(); return hipGetErrorString(e); }
int main() {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t GiB = 1ull << 30;
  for (int variant = 0; variant < 4; variant++) {
    const size_t align = variant == 1 ? GiB : (variant == 2 ? 4 * GiB : 0);
    const size_t chunk = variant == 3 ? GiB / 4 : GiB;
    void *va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, 64 * GiB, align, nullptr, 0);
    printf("== variant %d: reserve 64 GiB align %zu -> %s base %p; chunk %zu MiB\n", variant, align, E(e), va, chunk >> 20);
    if (e != hipSuccess) continue;
    std::vector<hipMemGenericAllocationHandle_t> hs;
    size_t off = 0;
    for (int i = 0; i < (variant == 3 ? 40 : 12); i++) {
      hipMemGenericAllocationHandle_t h;
      e = hipMemCreate(&h, chunk, &prop, 0);
      if (e != hipSuccess) { printf("  create #%d -> %s\n", i, E(e)); break; }
      hipError_t m = hipMemMap((char *)va + off, chunk, 0, h, 0);
      hipError_t a = m == hipSuccess ? hipMemSetAccess((char *)va + off, chunk, &acc, 1) : hipErrorUnknown;
      printf("  chunk #%d at +%.2f GiB: map %s, access %s\n", i, (double)off / GiB, E(m), m == hipSuccess ? E(a) : "-");
      if (m != hipSuccess) { (void)hipMemRelease(h); break; }
      hs.push_back(h);
      if (a != hipSuccess) { off += chunk; break; }
      off += chunk;
    }
    if (off) {
      hipError_t w = hipMemset((char *)va + off - 4096, 7, 4096);
      hipError_t s = hipDeviceSynchronize();
      printf("  touch last page at +%.2f GiB: %s / %s\n", (double)(off - 4096) / GiB, E(w), E(s));
    }
    size_t o = 0;
    for (auto h : hs) { (void)hipMemUnmap((char *)va + o, chunk); (void)hipMemRelease(h); o += chunk; }
    (void)hipMemAddressFree(va, 64 * GiB);
  }
  return 0;
}
