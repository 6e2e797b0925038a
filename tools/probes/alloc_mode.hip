// probe (r6): does the streaming rate of the RHS kernels' access shape depend on HOW the arrays were allocated?
// The same kernel as layout.hip mode 0 (24 arrays of 462400 x 80 doubles read one 128-byte piece per row at a time, one
// written) on arrays from (0) one hipMalloc each, (1) ONE hipMalloc arena, (2) the virtual-memory API: one physical handle
// per array (hipMemCreate at the recommended granularity), mapped into one reserved range, (3) the same with ONE physical
// handle for everything.  Every mode is allocated, timed and released several times: round 3 found that the physical pages
// an allocation happens to get move the RHS by +-3.5 % (profiles/r03_probe_placement.txt); the question here is whether an
// allocation mode gives the good placement every time.
//   hipcc -O3 --offload-arch=gfx950 tools/probes/alloc_mode.hip -o /tmp/alloc_mode && /tmp/alloc_mode [trials]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double dv2 __attribute__((ext_vector_type(2)));
struct Ptrs { const double *p[32]; };
__global__ void __launch_bounds__(256, 2) k(Ptrs P, int A, double *out, int rows, int K, int nchunk) {
   const int x = threadIdx.x, y = threadIdx.y;
   const int row = blockIdx.x * 32 + y;
   if (row >= rows) return;
   for (int c = 0; c < nchunk; ++c) {
      const size_t off = (size_t)row * K + c * 16 + x * 2;
      dv2 s = {0.0, 0.0};
      for (int a = 0; a < A; ++a) s += *reinterpret_cast<const dv2 *>(P.p[a] + off);
      __builtin_nontemporal_store(s, reinterpret_cast<dv2 *>(out + off));
   }
}
int main(int argc, char **argv) {
   const int rows = 462400, K = 80, nchunk = 5, A = 24, trials = argc > 1 ? atoi(argv[1]) : 4;
   const size_t n = (size_t)rows * K, bytes = n * 8;
   hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   hipMemAllocationProp prop{};
   prop.type = hipMemAllocationTypePinned;
   prop.location.type = hipMemLocationTypeDevice;
   prop.location.id = 0;
   size_t gran = 0;
   CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
   const size_t slot = (bytes + gran - 1) / gran * gran;
   printf("array %zu bytes, recommended granularity %zu, slot %zu\n", bytes, gran, slot);
   for (int mode = 0; mode < 4; ++mode) {
      for (int t = 0; t < trials; ++t) {
         Ptrs P{}; double *out = nullptr;
         std::vector<void *> frees; std::vector<hipMemGenericAllocationHandle_t> handles; void *range = nullptr; size_t rangeBytes = 0;
         void *perturb = nullptr; CK(hipMalloc(&perturb, (size_t)(1 + t) * 37 * 1024 * 1024)); // move the allocator between trials
         if (mode == 0) {
            for (int a = 0; a <= A; ++a) { void *p; CK(hipMalloc(&p, bytes)); frees.push_back(p); if (a < A) P.p[a] = (double *)p; else out = (double *)p; }
         } else if (mode == 1) {
            void *p; CK(hipMalloc(&p, slot * (A + 1))); frees.push_back(p);
            for (int a = 0; a <= A; ++a) { double *q = (double *)((char *)p + slot * a); if (a < A) P.p[a] = q; else out = q; }
         } else {
            rangeBytes = slot * (A + 1);
            CK(hipMemAddressReserve(&range, rangeBytes, 0, nullptr, 0));
            if (mode == 2) {
               for (int a = 0; a <= A; ++a) { hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, slot, &prop, 0)); handles.push_back(h);
                  CK(hipMemMap((char *)range + slot * a, slot, 0, h, 0)); }
            } else {
               hipMemGenericAllocationHandle_t h; CK(hipMemCreate(&h, rangeBytes, &prop, 0)); handles.push_back(h);
               CK(hipMemMap(range, rangeBytes, 0, h, 0));
            }
            hipMemAccessDesc acc{}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(range, rangeBytes, &acc, 1));
            for (int a = 0; a <= A; ++a) { double *q = (double *)((char *)range + slot * a); if (a < A) P.p[a] = q; else out = q; }
         }
         for (int a = 0; a < A; ++a) CK(hipMemset((void *)P.p[a], 0, bytes));
         CK(hipMemset(out, 0, bytes));
         float best = 1e9f;
         for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0));
            for (int it = 0; it < 5; ++it) hipLaunchKernelGGL(k, dim3((rows + 31) / 32), dim3(8, 32), 0, 0, P, A, out, rows, K, nchunk);
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5; if (ms < best) best = ms;
         }
         printf("mode %d (%s) trial %d: %.3f ms  %.2f TB/s\n", mode,
                mode == 0 ? "hipMalloc per array" : mode == 1 ? "one hipMalloc arena" : mode == 2 ? "VMM, one handle per array" : "VMM, one handle",
                t, best, (A + 1) * (double)bytes / (best * 1e-3) / 1e12);
         CK(hipDeviceSynchronize());
         if (range) { CK(hipMemUnmap(range, rangeBytes)); for (auto h : handles) CK(hipMemRelease(h)); CK(hipMemAddressFree(range, rangeBytes)); }
         for (void *p : frees) CK(hipFree(p));
         CK(hipFree(perturb));
      }
   }
   return 0;
}
