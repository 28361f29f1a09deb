// diagnostic: which hipMemMap shapes does this driver accept?  (scripts/diag, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); printf("%-60s -> %s\n", #x, hipGetErrorString(e)); if (e != hipSuccess) (void)hipGetLastError(); } while (0)
int main() {
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = 0;
  size_t gran = 0, gmin = 0;
  CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
  CK(hipMemGetAllocationGranularity(&gmin, &prop, hipMemAllocationGranularityMinimum));
  printf("granularity recommended %zu minimum %zu\n", gran, gmin);
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  const size_t MiB = 1ull << 20, GiB = 1ull << 30;
  for (size_t big : {1 * GiB, 2 * GiB, 3 * GiB + 768 * MiB, 4 * GiB, 8 * GiB, 30 * GiB}) {
    printf("== one chunk of %.2f GiB behind a 512 MiB chunk, 64 GiB reservation\n", (double)big / GiB);
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, 64 * GiB, 0, nullptr, 0));
    hipMemGenericAllocationHandle_t h0, h1;
    CK(hipMemCreate(&h0, 512 * MiB, &prop, 0));
    CK(hipMemMap(va, 512 * MiB, 0, h0, 0));
    CK(hipMemSetAccess(va, 512 * MiB, &acc, 1));
    hipError_t e = hipMemCreate(&h1, big, &prop, 0);
    printf("hipMemCreate(big) -> %s\n", hipGetErrorString(e));
    if (e == hipSuccess) {
      CK(hipMemMap((char *)va + 512 * MiB, big, 0, h1, 0));
      CK(hipMemSetAccess((char *)va + 512 * MiB, big, &acc, 1));
      CK(hipMemset((char *)va + 512 * MiB + big - 4096, 1, 4096));
      CK(hipDeviceSynchronize());
      // re-map both into a new range
      void *vb = nullptr;
      CK(hipMemAddressReserve(&vb, 128 * GiB, 0, nullptr, 0));
      CK(hipMemUnmap(va, 512 * MiB));
      CK(hipMemUnmap((char *)va + 512 * MiB, big));
      CK(hipMemMap(vb, 512 * MiB, 0, h0, 0));
      CK(hipMemMap((char *)vb + 512 * MiB, big, 0, h1, 0));
      CK(hipMemSetAccess(vb, 512 * MiB + big, &acc, 1));
      CK(hipMemset((char *)vb + 512 * MiB + big - 4096, 2, 4096));
      CK(hipDeviceSynchronize());
      CK(hipMemUnmap(vb, 512 * MiB));
      CK(hipMemUnmap((char *)vb + 512 * MiB, big));
      CK(hipMemAddressFree(vb, 128 * GiB));
      CK(hipMemRelease(h1));
    } else {
      (void)hipGetLastError();
      CK(hipMemUnmap(va, 512 * MiB));
    }
    CK(hipMemRelease(h0));
    CK(hipMemAddressFree(va, 64 * GiB));
  }
  return 0;
}
