// probe: HBM rate of the RHS kernels' access pattern ([rows][K] arrays read one 128-byte level chunk per row at a
// time, 32 consecutive rows per workgroup) against a level-chunk-major layout ([chunk][rows][16 levels]) where the
// same workgroup reads 4 KiB contiguous pieces.  A arrays in, 1 array out, nothing else.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double dv2 __attribute__((ext_vector_type(2)));
struct Ptrs { const double *p[32]; };
template <int MODE>
__global__ void __launch_bounds__(256, 2) k(Ptrs P, int A, double *out, int rows, int K, int nchunk) {
   const int x = threadIdx.x, y = threadIdx.y;
   const int row = blockIdx.x * 32 + y;
   if (row >= rows) return;
   for (int c = 0; c < nchunk; ++c) {
      size_t off;
      if (MODE == 0) off = (size_t)row * K + c * 16 + x * 2;                 // [rows][K]
      else           off = ((size_t)c * rows + row) * 16 + x * 2;            // [chunk][rows][16]
      dv2 s = {0.0, 0.0};
      for (int a = 0; a < A; ++a) s += *reinterpret_cast<const dv2 *>(P.p[a] + off);
      __builtin_nontemporal_store(s, reinterpret_cast<dv2 *>(out + off));
   }
}
// mode 2 (r6): [rows][K] again, but the workgroup's threads cover ALL level chunks of its rows at once -- (8 lanes, 5 chunks,
// 6 rows) = 240 threads, one 128-byte piece each, no chunk loop -- so every 640-byte row is asked for as one contiguous piece
// within one instruction issue instead of as five pieces a chunk-loop iteration apart (DRAM page locality of the access shape).
__global__ void __launch_bounds__(960) kRow(Ptrs P, int A, double *out, int rows, int K, int rowsPerWG) {
   const int x = threadIdx.x, c = threadIdx.y;
   const int row = blockIdx.x * rowsPerWG + threadIdx.z;
   if (row >= rows) return;
   const size_t off = (size_t)row * K + c * 16 + x * 2;
   dv2 s = {0.0, 0.0};
   for (int a = 0; a < A; ++a) s += *reinterpret_cast<const dv2 *>(P.p[a] + off);
   __builtin_nontemporal_store(s, reinterpret_cast<dv2 *>(out + off));
}
int main(int argc, char **argv) {
   const int rows = 462400, K = 80, nchunk = 5, A = argc > 1 ? atoi(argv[1]) : 24;
   const size_t n = (size_t)rows * K;
   Ptrs P{};
   for (int a = 0; a < A; ++a) { double *p; hipMalloc(&p, n * 8); hipMemset(p, 0, n * 8); P.p[a] = p; }
   double *out; hipMalloc(&out, n * 8);
   hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
   for (int mode = 0; mode < 2; ++mode) {
      for (int rep = 0; rep < 3; ++rep) {
         hipEventRecord(e0);
         for (int it = 0; it < 5; ++it) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3((rows + 31) / 32), dim3(8, 32), 0, 0, P, A, out, rows, K, nchunk);
            else           hipLaunchKernelGGL(k<1>, dim3((rows + 31) / 32), dim3(8, 32), 0, 0, P, A, out, rows, K, nchunk);
         }
         hipEventRecord(e1); hipEventSynchronize(e1);
         float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
         printf("mode %d (%s) A=%d: %.3f ms  %.2f TB/s\n", mode, mode ? "[chunk][rows][16]" : "[rows][K], 128 B per row at a time",
                A, ms, (A + 1) * n * 8 / (ms * 1e-3) / 1e12);
      }
   }
   for (int rpw : {6, 12, 24}) {   // 240 / 480 / 960 threads per workgroup
      for (int rep = 0; rep < 3; ++rep) {
         hipEventRecord(e0);
         for (int it = 0; it < 5; ++it)
            hipLaunchKernelGGL(kRow, dim3((rows + rpw - 1) / rpw), dim3(8, 5, rpw), 0, 0, P, A, out, rows, K, rpw);
         hipEventRecord(e1); hipEventSynchronize(e1);
         float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
         printf("mode 2 ([rows][K], whole 640-byte rows at once, %d rows per workgroup) A=%d: %.3f ms  %.2f TB/s\n", rpw, A, ms,
                (A + 1) * n * 8 / (ms * 1e-3) / 1e12);
      }
   }
   return 0;
}
