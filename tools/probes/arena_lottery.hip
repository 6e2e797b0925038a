// probe (r6): the placement lottery, held open.  N arenas of 25 x 296 MB are allocated ONE AFTER THE OTHER AND KEPT (so
// each gets different physical memory), every one is timed with the RHS kernels' access shape (24 planes read one 128-byte
// piece per row at a time, one written) twice, then all are timed again in reverse order: is the rate a property of the
// arena (stable across re-timing), how wide is the spread, and how often does a fresh arena land in the fast group?
//   hipcc -O3 --offload-arch=gfx950 tools/probes/arena_lottery.hip -o /tmp/arena_lottery && /tmp/arena_lottery [arenas]
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
static const int rows = 462400, K = 80, nchunk = 5, A = 24;
static hipEvent_t e0, e1;
static double rate(char *base, size_t slot) {
   Ptrs P{};
   for (int a = 0; a < A; ++a) P.p[a] = (double *)(base + slot * a);
   double *out = (double *)(base + slot * A);
   const size_t bytes = (size_t)rows * K * 8;
   float best = 1e9f;
   for (int rep = 0; rep < 3; ++rep) {
      CK(hipEventRecord(e0));
      for (int it = 0; it < 4; ++it) hipLaunchKernelGGL(k, dim3((rows + 31) / 32), dim3(8, 32), 0, 0, P, A, out, rows, K, nchunk);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 4; if (ms < best) best = ms;
   }
   return (A + 1) * (double)bytes / (best * 1e-3) / 1e12;
}
int main(int argc, char **argv) {
   const int N = argc > 1 ? atoi(argv[1]) : 8;
   const size_t bytes = (size_t)rows * K * 8, slot = (bytes + 4095) / 4096 * 4096, arena = slot * (A + 1);
   CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
   std::vector<char *> As;
   for (int i = 0; i < N; ++i) {
      void *p; CK(hipMalloc(&p, arena)); CK(hipMemset(p, 0, arena)); CK(hipDeviceSynchronize());
      As.push_back((char *)p);
      printf("arena %d at %p: %.2f  %.2f TB/s\n", i, p, rate((char *)p, slot), rate((char *)p, slot));
   }
   for (int i = N - 1; i >= 0; --i) printf("arena %d again: %.2f TB/s\n", i, rate(As[i], slot));
   // the same physical arenas with the planes SKEWED inside them (plane a starts a * 4352 bytes later): virtual = physical
   // offsets inside one allocation's contiguous pieces
   for (int i = 0; i < N; ++i) printf("arena %d, planes at a smaller stride (slot - 1 MiB): %.2f TB/s\n", i, rate(As[i], slot - (1 << 20)));
   for (char *p : As) CK(hipFree(p));
   return 0;
}
