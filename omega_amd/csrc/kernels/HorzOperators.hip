// HorzOperators.hip -- the reference's reusable horizontal operators as HIP kernels
// (components/omega/src/ocn/HorzOperators.h:9-187): DivergenceOnCell, GradientOnEdge, CurlOnVertex,
// TangentialReconOnEdge, InterpCellToEdge.  Each body keeps the functor's operation order (the
// level-independent prefix of a product chain is evaluated once per element slot while the tile is
// staged into LDS, with the same operations in the same order), so results equal the CPU functors
// bit for bit.  Compiled with -ffp-contract=off.  Same tile skeleton as the RHS kernels: vertical
// index innermost, connectivity + coefficients of the tile staged in LDS, one 16-byte access per lane.
#include "KernelCommon.h"
#include "Kernels.h"

namespace OMEGA {

// DivergenceOnCell (HorzOperators.h:13-33):
//   DivCell(i,k) = - sum_j DvEdge(e_j) * EdgeSignOnCell(i,j) * VecEdge(e_j,k) * (1/AreaCell(i))
struct DivergenceOnCellBody {
   MeshView M;
   int K;
   const Real *VecEdge;
   Real *DivCell;
   struct Lds {
      Real *DvS, *InvA;
      int *Edge, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME = M.MaxEdges;
      return ldsRound8(sizeof(Real) * Tile * ME) + ldsRound8(sizeof(Real) * Tile) + ldsRound8(sizeof(int) * Tile * ME) +
             ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      const int ME = M.MaxEdges;
      LdsCarver C{P};
      Lds L;
      L.DvS  = C.take<Real>(Tile * ME);
      L.InvA = C.take<Real>(Tile);
      L.Edge = C.take<int>(Tile * ME);
      L.N    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME = M.MaxEdges;
      for (int I = Tid; I < Cnt * ME; I += NThr) {
         const size_t G = (size_t)First * ME + I;
         const int E    = M.EdgesOnCell[G];
         L.Edge[I]      = E;
         L.DvS[I]       = M.DvEdge[E] * M.EdgeSignOnCell[G];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         L.InvA[I] = 1. / M.AreaCell[First + I];
         L.N[I]    = M.NEdgesOnCell[First + I];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int ICell, int Kv) const {
      const int ME    = M.MaxEdges;
      const Real InvA = L.InvA[Le];
      T Tmp           = splat<T>(0.0);
      const int N     = L.N[Le];
      for (int J = 0; J < N; ++J)
         Tmp -= L.DvS[Le * ME + J] * ldk<T>(VecEdge, L.Edge[Le * ME + J], K, Kv) * InvA;
      stk<T>(DivCell, ICell, K, Kv, Tmp);
   }
};

// GradientOnEdge (HorzOperators.h:47-60): GradEdge(e,k) = (1/DcEdge(e)) * (Scalar(c1,k) - Scalar(c0,k))
struct GradientOnEdgeBody {
   MeshView M;
   int K;
   const Real *ScalarCell;
   Real *GradEdge;
   struct Lds {
      Real *InvDc;
      int *C0, *C1;
   };
   size_t ldsBytes(int Tile) const { return ldsRound8(sizeof(Real) * Tile) + ldsRound8(sizeof(int) * Tile) * 2; }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      LdsCarver C{P};
      Lds L;
      L.InvDc = C.take<Real>(Tile);
      L.C0    = C.take<int>(Tile);
      L.C1    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt; I += NThr) {
         const int E = First + I;
         L.InvDc[I]  = 1. / M.DcEdge[E];
         L.C0[I]     = M.CellsOnEdge[2 * E];
         L.C1[I]     = M.CellsOnEdge[2 * E + 1];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IEdge, int Kv) const {
      const T G = L.InvDc[Le] * (ldk<T>(ScalarCell, L.C1[Le], K, Kv) - ldk<T>(ScalarCell, L.C0[Le], K, Kv));
      stk<T>(GradEdge, IEdge, K, Kv, G);
   }
};

// CurlOnVertex (HorzOperators.h:71-93):
//   CurlVertex(v,k) = sum_j DcEdge(e_j) * EdgeSignOnVertex(v,j) * VecEdge(e_j,k) * (1/AreaTriangle(v))
struct CurlOnVertexBody {
   MeshView M;
   int K;
   const Real *VecEdge;
   Real *CurlVertex;
   struct Lds {
      Real *DcS, *InvA;
      int *Edge;
   };
   size_t ldsBytes(int Tile) const {
      const int VD = M.VertexDegree;
      return ldsRound8(sizeof(Real) * Tile * VD) + ldsRound8(sizeof(Real) * Tile) + ldsRound8(sizeof(int) * Tile * VD);
   }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      const int VD = M.VertexDegree;
      LdsCarver C{P};
      Lds L;
      L.DcS  = C.take<Real>(Tile * VD);
      L.InvA = C.take<Real>(Tile);
      L.Edge = C.take<int>(Tile * VD);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int VD = M.VertexDegree;
      for (int I = Tid; I < Cnt * VD; I += NThr) {
         const size_t G = (size_t)First * VD + I;
         const int E    = M.EdgesOnVertex[G];
         L.Edge[I]      = E;
         L.DcS[I]       = M.DcEdge[E] * M.EdgeSignOnVertex[G];
      }
      for (int I = Tid; I < Cnt; I += NThr)
         L.InvA[I] = 1. / M.AreaTriangle[First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IVertex, int Kv) const {
      const int VD    = M.VertexDegree;
      const Real InvA = L.InvA[Le];
      T Tmp           = splat<T>(0.0);
      for (int J = 0; J < VD; ++J)
         Tmp += L.DcS[Le * VD + J] * ldk<T>(VecEdge, L.Edge[Le * VD + J], K, Kv) * InvA;
      stk<T>(CurlVertex, IVertex, K, Kv, Tmp);
   }
};

// TangentialReconOnEdge (HorzOperators.h:107-126): ReconEdge(e,k) = sum_j WeightsOnEdge(e,j) * VecEdge(e'_j,k)
struct TangentialReconOnEdgeBody {
   MeshView M;
   int K;
   const Real *VecEdge;
   Real *ReconEdge;
   struct Lds {
      Real *W;
      int *EoE, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME2 = M.MaxEdges2;
      return ldsRound8(sizeof(Real) * Tile * ME2) + ldsRound8(sizeof(int) * Tile * ME2) + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      const int ME2 = M.MaxEdges2;
      LdsCarver C{P};
      Lds L;
      L.W   = C.take<Real>(Tile * ME2);
      L.EoE = C.take<int>(Tile * ME2);
      L.N   = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME2 = M.MaxEdges2;
      for (int I = Tid; I < Cnt * ME2; I += NThr) {
         const size_t G = (size_t)First * ME2 + I;
         L.W[I]         = M.WeightsOnEdge[G];
         L.EoE[I]       = M.EdgesOnEdge[G];
      }
      for (int I = Tid; I < Cnt; I += NThr)
         L.N[I] = M.NEdgesOnEdge[First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IEdge, int Kv) const {
      const int ME2 = M.MaxEdges2;
      T Tmp         = splat<T>(0.0);
      const int N   = L.N[Le];
      for (int J = 0; J < N; ++J)
         Tmp += L.W[Le * ME2 + J] * ldk<T>(VecEdge, L.EoE[Le * ME2 + J], K, Kv);
      stk<T>(ReconEdge, IEdge, K, Kv, Tmp);
   }
};

void launchDivergenceOnCell(const MeshView &M, int N, int K, int Pitch, Real *DivCell, const Real *VecEdge,
                            hipStream_t S) {
   launchTile(DivergenceOnCellBody{M, K, VecEdge, DivCell}, N, K, S, Pitch);
}
void launchGradientOnEdge(const MeshView &M, int N, int K, int Pitch, Real *GradEdge, const Real *ScalarCell,
                          hipStream_t S) {
   launchTile(GradientOnEdgeBody{M, K, ScalarCell, GradEdge}, N, K, S, Pitch);
}
void launchCurlOnVertex(const MeshView &M, int N, int K, int Pitch, Real *CurlVertex, const Real *VecEdge,
                        hipStream_t S) {
   launchTile(CurlOnVertexBody{M, K, VecEdge, CurlVertex}, N, K, S, Pitch);
}
void launchTangentialReconOnEdge(const MeshView &M, int N, int K, int Pitch, Real *ReconEdge, const Real *VecEdge,
                                 hipStream_t S) {
   launchTile(TangentialReconOnEdgeBody{M, K, VecEdge, ReconEdge}, N, K, S, Pitch);
}

// InterpCellToEdge on a 1-D cell array (HorzOperators.h:137-187): anisotropic = mean of the two cells of the
// edge; isotropic = kite-area weighted mean over the cells of its two vertices
__global__ void interpCellToEdgeKernel(MeshView M, int N, const Real *ArrayCell, Real *ArrayEdge, int Isotropic) {
   const int IEdge = blockIdx.x * blockDim.x + threadIdx.x;
   if (IEdge >= N)
      return;
   if (!Isotropic) { // :153-159
      const int JCell0 = M.CellsOnEdge[IEdge * 2 + 0], JCell1 = M.CellsOnEdge[IEdge * 2 + 1];
      ArrayEdge[IEdge] = 0.5 * (ArrayCell[JCell0] + ArrayCell[JCell1]);
      return;
   }
   const int VD = M.VertexDegree; // :161-180
   Real Accum = 0, AreaAccum = 0;
   for (int J = 0; J < 2; ++J) {
      const int JVertex = M.VerticesOnEdge[IEdge * 2 + J];
      for (int L = 0; L < VD; ++L) {
         const Real KiteArea = M.KiteAreasOnVertex[JVertex * VD + L];
         const int LCell     = M.CellsOnVertex[JVertex * VD + L];
         Accum += ArrayCell[LCell] * KiteArea;
         AreaAccum += KiteArea;
      }
   }
   const Real InvAreaAccum = 1. / AreaAccum;
   ArrayEdge[IEdge]        = Accum * InvAreaAccum;
}
void launchInterpCellToEdge(const MeshView &M, int N, Real *ArrayEdge, const Real *ArrayCell, int Isotropic,
                            hipStream_t S) {
   if (N <= 0)
      return;
   hipLaunchKernelGGL(interpCellToEdgeKernel, dim3((N + 255) / 256), dim3(256), 0, S, M, N, ArrayCell, ArrayEdge,
                      Isotropic);
   HIP_CHECK(hipGetLastError());
}

} // namespace OMEGA
