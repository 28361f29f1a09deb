// diagnostic 3: the exact sequence that fails inside the library: reservation of 8.25 GiB, 4 x 1 GiB chunks, then a
// 256 MiB chunk at +4 GiB -- alone, with a second live reservation, and with the reservation rounded to 1 / 4 GiB.
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
  const size_t GiB = 1ull << 30, MiB = 1ull << 20;
  const size_t sizes[] = {8858370048ull, 9 * GiB, 12 * GiB, 8858370048ull};
  for (int variant = 0; variant < 4; variant++) {
    void *other = nullptr;
    hipMemGenericAllocationHandle_t oh;
    if (variant == 3) {  // a second, older reservation with one chunk alive (the buffer being rebuilt)
      printf("other reserve: %s\n", E(hipMemAddressReserve(&other, 2 * GiB, 0, nullptr, 0)));
      printf("other create/map/access: %s %s %s\n", E(hipMemCreate(&oh, 512 * MiB, &prop, 0)), E(hipMemMap(other, 512 * MiB, 0, oh, 0)),
             E(hipMemSetAccess(other, 512 * MiB, &acc, 1)));
    }
    void *va = nullptr;
    hipError_t e = hipMemAddressReserve(&va, sizes[variant], 0, nullptr, 0);
    printf("== variant %d: reserve %zu -> %s base %p\n", variant, sizes[variant], E(e), va);
    size_t off = 0;
    const size_t chunks[] = {GiB, GiB, GiB, GiB, 256 * MiB, 256 * MiB, GiB};
    for (size_t c : chunks) {
      hipMemGenericAllocationHandle_t h;
      hipError_t cr = hipMemCreate(&h, c, &prop, 0);
      hipError_t m = cr == hipSuccess ? hipMemMap((char *)va + off, c, 0, h, 0) : hipErrorUnknown;
      hipError_t a = m == hipSuccess ? hipMemSetAccess((char *)va + off, c, &acc, 1) : hipErrorUnknown;
      printf("  %4zu MiB at +%.2f GiB: create %s, map %s, access %s\n", c >> 20, (double)off / GiB, E(cr), E(m), E(a));
      if (m != hipSuccess || a != hipSuccess) break;
      off += c;
    }
  }
  return 0;
}
