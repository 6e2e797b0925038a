// AuxKernels.hip -- one HIP kernel per AuxiliaryState launch of the reference
// (components/omega/src/ocn/AuxiliaryState.cpp:79-182).  Each kernel body restates the
// reference functor (file:line given at each body) with the level-independent prefix of
// every product chain taken from the host-built coefficient tables (HorzMesh.h), so the
// floating-point operation order -- and hence every bit of the result -- is the
// reference's.  Compiled with -ffp-contract=off.
#include "KernelCommon.h"
#include "Kernels.h"

namespace OMEGA {

// ---------------------------------------------------------------------------------------
// vertexAuxState1: VorticityAuxVars::computeVarsOnVertex (auxiliaryVars/VorticityAuxVars.h:24-59)
struct VortVertexBody {
   MeshView M;
   int K;
   const Real *H, *U;
   Real *RelVort, *NormRelVort, *NormPlanetVort, *InvThick;
   int StoreNorm, StoreInv;
   const int *List = nullptr; // optional vertex list (MeshView::OrphanVertices)
   struct Lds {
      Real *KiteC, *VortC, *F;
      int *Cell, *Edge;
   };
   size_t ldsBytes(int Tile) const {
      const int VD = M.VertexDegree;
      return ldsRound8(sizeof(Real) * Tile * VD) * 2 + ldsRound8(sizeof(Real) * Tile) +
             ldsRound8(sizeof(int) * Tile * VD) * 2;
   }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      const int VD = M.VertexDegree;
      LdsCarver C{P};
      Lds L;
      L.KiteC = C.take<Real>(Tile * VD);
      L.VortC = C.take<Real>(Tile * VD);
      L.F     = C.take<Real>(Tile);
      L.Cell  = C.take<int>(Tile * VD);
      L.Edge  = C.take<int>(Tile * VD);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int VD = M.VertexDegree;
      for (int I = Tid; I < Cnt * VD; I += NThr) {
         size_t G = (size_t)First * VD + I;
         if (List) {
            const int Vl = I / VD;
            G            = (size_t)List[First + Vl] * VD + (I - Vl * VD);
         }
         L.KiteC[I]     = M.KiteCoefOnVertex[G];
         L.VortC[I]     = M.VortCoefOnVertex[G];
         L.Cell[I]      = M.CellsOnVertex[G];
         L.Edge[I]      = M.EdgesOnVertex[G];
      }
      for (int I = Tid; I < Cnt; I += NThr)
         L.F[I] = M.FVertex[List ? List[First + I] : First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      const int VD      = M.VertexDegree;
      const int IVertex = List ? List[IElem] : IElem;
      T LayerThickVertex = splat<T>(0.0), RelVortTmp = splat<T>(0.0);
      for (int J = 0; J < VD; ++J) {
         const T Hc = ldk<T>(H, L.Cell[Le * VD + J], K, Kv);
         const T Ue = ldk<T>(U, L.Edge[Le * VD + J], K, Kv);
         LayerThickVertex += L.KiteC[Le * VD + J] * Hc;
         RelVortTmp += L.VortC[Le * VD + J] * Ue;
      }
      const T Inv = 1. / LayerThickVertex;
      stk<T>(RelVort, IVertex, K, Kv, RelVortTmp);
      if (StoreNorm) {
         stk<T>(NormRelVort, IVertex, K, Kv, RelVortTmp * Inv);
         stk<T>(NormPlanetVort, IVertex, K, Kv, L.F[Le] * Inv);
      }
      if (StoreInv)
         stk<T>(InvThick, IVertex, K, Kv, Inv);
   }
};

void launchVertexAuxState1(const MeshView &M, int K, const AuxPtrs &A, const Real *H, const Real *U, hipStream_t S,
                           bool StoreNorm, bool StoreInv) {
   VortVertexBody B{M,
                    K,
                    H,
                    U,
                    A.RelVortVertex,
                    A.NormRelVortVertex,
                    A.NormPlanetVortVertex,
                    A.InvThickVertex,
                    StoreNorm ? 1 : 0,
                    StoreInv ? 1 : 0};
   launchTile(B, M.NVerticesAll, K, S);
}
void launchVertexAuxState1List(const MeshView &M, int K, const AuxPtrs &A, const Real *H, const Real *U, hipStream_t S,
                               const I4 *Vertices, int N) {
   if (N <= 0)
      return;
   VortVertexBody B{M, K, H, U, A.RelVortVertex, A.NormRelVortVertex, A.NormPlanetVortVertex, A.InvThickVertex, 0, 1, Vertices};
   launchTile(B, N, K, S);
}

// ---------------------------------------------------------------------------------------
// cellAuxState1: KineticAuxVars::computeVarsOnCell (auxiliaryVars/KineticAuxVars.h:20-47)
struct KineticCellBody {
   MeshView M;
   int K;
   const Real *U;
   Real *KE, *Div;
   struct Lds {
      Real *KEC, *DivC;
      int *Edge, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME = M.MaxEdges;
      return ldsRound8(sizeof(Real) * Tile * ME) * 2 + ldsRound8(sizeof(int) * Tile * ME) +
             ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      const int ME = M.MaxEdges;
      LdsCarver C{P};
      Lds L;
      L.KEC  = C.take<Real>(Tile * ME);
      L.DivC = C.take<Real>(Tile * ME);
      L.Edge = C.take<int>(Tile * ME);
      L.N    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME = M.MaxEdges;
      for (int I = Tid; I < Cnt * ME; I += NThr) {
         const size_t G = (size_t)First * ME + I;
         L.KEC[I]       = M.KECoefOnCell[G];
         L.DivC[I]      = M.DivCoefOnCell[G];
         L.Edge[I]      = M.EdgesOnCell[G];
      }
      for (int I = Tid; I < Cnt; I += NThr)
         L.N[I] = M.NEdgesOnCell[First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int ICell, int Kv) const {
      const int ME = M.MaxEdges;
      T KETmp = splat<T>(0.0), DivTmp = splat<T>(0.0);
      const int N = L.N[Le];
      for (int J = 0; J < N; ++J) {
         const T Ue = ldk<T>(U, L.Edge[Le * ME + J], K, Kv);
         KETmp += L.KEC[Le * ME + J] * Ue * Ue;
         DivTmp -= L.DivC[Le * ME + J] * Ue;
      }
      stk<T>(KE, ICell, K, Kv, KETmp);
      stk<T>(Div, ICell, K, Kv, DivTmp);
   }
};

void launchCellAuxState1(const MeshView &M, int K, const AuxPtrs &A, const Real *U, hipStream_t S) {
   KineticCellBody B{M, K, U, A.KineticEnergyCell, A.VelocityDivCell};
   launchTile(B, M.NCellsAll, K, S);
}

// ---------------------------------------------------------------------------------------
// edgeAuxState1: WindForcingAuxVars::computeVarsOnEdge (auxiliaryVars/WindForcingAuxVars.h:22-29)
// with InterpCellToEdge (HorzOperators.h:137-187).  1-D (no vertical index).
__device__ inline Real interpCellToEdge(const MeshView &M, int IEdge, const Real *ArrayCell, int Isotropic) {
   if (!Isotropic) {
      const int JCell0 = M.CellsOnEdge[IEdge * 2 + 0], JCell1 = M.CellsOnEdge[IEdge * 2 + 1];
      return 0.5 * (ArrayCell[JCell0] + ArrayCell[JCell1]);
   }
   const int VD = M.VertexDegree;
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
   return Accum * InvAreaAccum;
}

__global__ void windEdgeKernel(MeshView M, const Real *Zonal, const Real *Merid, Real *NormalStress, int Isotropic) {
   const int IEdge = blockIdx.x * blockDim.x + threadIdx.x;
   if (IEdge >= M.NEdgesAll)
      return;
   const Real ZonalStressEdge = interpCellToEdge(M, IEdge, Zonal, Isotropic);
   const Real MeridStressEdge = interpCellToEdge(M, IEdge, Merid, Isotropic);
   NormalStress[IEdge] = cos(M.AngleEdge[IEdge]) * ZonalStressEdge + sin(M.AngleEdge[IEdge]) * MeridStressEdge;
}

void launchEdgeAuxState1(const MeshView &M, const AuxPtrs &A, int Isotropic, hipStream_t S) {
   if (M.NEdgesAll <= 0)
      return;
   hipLaunchKernelGGL(windEdgeKernel, dim3((M.NEdgesAll + 255) / 256), dim3(256), 0, S, M, A.ZonalStressCell,
                      A.MeridStressCell, A.NormalStressEdge, Isotropic);
}

// ---------------------------------------------------------------------------------------
// edgeAuxState2: VorticityAuxVars::computeVarsOnEdge (VorticityAuxVars.h:61-76),
// LayerThicknessAuxVars::computeVarsOnEdge (LayerThicknessAuxVars.h:25-61),
// VelocityDel2AuxVars::computeVarsOnEdge (VelocityDel2AuxVars.h:21-45)
template <bool DoVort, bool DoDel2> struct EdgeAux2Body {
   MeshView M;
   int K;
   const Real *H, *U;
   AuxPtrs A;
   int FluxUpwind;
   struct Lds {
      Real *InvDc, *InvDv2, *Mask;
      int *C0, *C1, *V0, *V1;
   };
   size_t ldsBytes(int Tile) const { return ldsRound8(sizeof(Real) * Tile) * 3 + ldsRound8(sizeof(int) * Tile) * 4; }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      LdsCarver C{P};
      Lds L;
      L.InvDc  = C.take<Real>(Tile);
      L.InvDv2 = C.take<Real>(Tile);
      L.Mask   = C.take<Real>(Tile);
      L.C0     = C.take<int>(Tile);
      L.C1     = C.take<int>(Tile);
      L.V0     = C.take<int>(Tile);
      L.V1     = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt; I += NThr) {
         const int E = First + I;
         L.InvDc[I]  = M.InvDcEdge[E];
         L.InvDv2[I] = M.InvDvEdgeDel2[E];
         L.Mask[I]   = M.EdgeMask1D[E];
         L.C0[I]     = M.CellsOnEdge[2 * E];
         L.C1[I]     = M.CellsOnEdge[2 * E + 1];
         L.V0[I]     = M.VerticesOnEdge[2 * E];
         L.V1[I]     = M.VerticesOnEdge[2 * E + 1];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IEdge, int Kv) const {
      const int C0 = L.C0[Le], C1 = L.C1[Le];
      if (DoVort) {
         const int V0 = L.V0[Le], V1 = L.V1[Le];
         stk<T>(A.NormRelVortEdge, IEdge, K, Kv,
                0.5 * (ldk<T>(A.NormRelVortVertex, V0, K, Kv) + ldk<T>(A.NormRelVortVertex, V1, K, Kv)));
         stk<T>(A.NormPlanetVortEdge, IEdge, K, Kv,
                0.5 * (ldk<T>(A.NormPlanetVortVertex, V0, K, Kv) + ldk<T>(A.NormPlanetVortVertex, V1, K, Kv)));
      }
      const T H0 = ldk<T>(H, C0, K, Kv), H1 = ldk<T>(H, C1, K, Kv);
      const T Mean = 0.5 * (H0 + H1);
      stk<T>(A.MeanLayerThickEdge, IEdge, K, Kv, Mean);
      if (!FluxUpwind) {
         stk<T>(A.FluxLayerThickEdge, IEdge, K, Kv, Mean);
      } else {
         const T Ue = ldk<T>(U, IEdge, K, Kv);
         stk<T>(A.FluxLayerThickEdge, IEdge, K, Kv, upwind(Ue, H0, H1));
      }
      if (DoDel2) {
         const int V0 = L.V0[Le], V1 = L.V1[Le];
         const T GradDiv =
             (ldk<T>(A.VelocityDivCell, C1, K, Kv) - ldk<T>(A.VelocityDivCell, C0, K, Kv)) * L.InvDc[Le];
         const T CurlVort =
             -(ldk<T>(A.RelVortVertex, V1, K, Kv) - ldk<T>(A.RelVortVertex, V0, K, Kv)) * L.InvDv2[Le];
         stk<T>(A.Del2Edge, IEdge, K, Kv, L.Mask[Le] * GradDiv + CurlVort);
      }
   }
};

void launchEdgeAuxState2(const MeshView &M, int K, const AuxPtrs &A, const Real *H, const Real *U, int FluxUpwind,
                         hipStream_t S) {
   EdgeAux2Body<true, true> B{M, K, H, U, A, FluxUpwind};
   launchTile(B, M.NEdgesAll, K, S);
}
void launchLayerThickAuxEdge(const MeshView &M, int K, const AuxPtrs &A, const Real *H, const Real *U, int FluxUpwind,
                             hipStream_t S) {
   EdgeAux2Body<false, false> B{M, K, H, U, A, FluxUpwind};
   launchTile(B, M.NEdgesAll, K, S);
}

// ---------------------------------------------------------------------------------------
// vertexAuxState2: VelocityDel2AuxVars::computeVarsOnVertex (VelocityDel2AuxVars.h:69-89)
struct Del2VertexBody {
   MeshView M;
   int K;
   const Real *Del2Edge;
   Real *Del2RelVort;
   struct Lds {
      Real *VortC;
      int *Edge;
   };
   size_t ldsBytes(int Tile) const {
      const int VD = M.VertexDegree;
      return ldsRound8(sizeof(Real) * Tile * VD) + ldsRound8(sizeof(int) * Tile * VD);
   }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      const int VD = M.VertexDegree;
      LdsCarver C{P};
      Lds L;
      L.VortC = C.take<Real>(Tile * VD);
      L.Edge  = C.take<int>(Tile * VD);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int VD = M.VertexDegree;
      for (int I = Tid; I < Cnt * VD; I += NThr) {
         const size_t G = (size_t)First * VD + I;
         L.VortC[I]     = M.VortCoefOnVertex[G];
         L.Edge[I]      = M.EdgesOnVertex[G];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IVertex, int Kv) const {
      const int VD = M.VertexDegree;
      T Tmp = splat<T>(0.0);
      for (int J = 0; J < VD; ++J)
         Tmp += L.VortC[Le * VD + J] * ldk<T>(Del2Edge, L.Edge[Le * VD + J], K, Kv);
      stk<T>(Del2RelVort, IVertex, K, Kv, Tmp);
   }
};
void launchVertexAuxState2(const MeshView &M, int K, const AuxPtrs &A, hipStream_t S) {
   Del2VertexBody B{M, K, A.Del2Edge, A.Del2RelVortVertex};
   launchTile(B, M.NVerticesAll, K, S);
}

// ---------------------------------------------------------------------------------------
// cellAuxState2: VelocityDel2AuxVars::computeVarsOnCell (VelocityDel2AuxVars.h:47-67)
struct Del2CellBody {
   MeshView M;
   int K;
   const Real *Del2Edge;
   Real *Del2Div;
   struct Lds {
      Real *DivC;
      int *Edge, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME = M.MaxEdges;
      return ldsRound8(sizeof(Real) * Tile * ME) + ldsRound8(sizeof(int) * Tile * ME) + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      const int ME = M.MaxEdges;
      LdsCarver C{P};
      Lds L;
      L.DivC = C.take<Real>(Tile * ME);
      L.Edge = C.take<int>(Tile * ME);
      L.N    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME = M.MaxEdges;
      for (int I = Tid; I < Cnt * ME; I += NThr) {
         const size_t G = (size_t)First * ME + I;
         L.DivC[I]      = M.DivCoefOnCell[G];
         L.Edge[I]      = M.EdgesOnCell[G];
      }
      for (int I = Tid; I < Cnt; I += NThr)
         L.N[I] = M.NEdgesOnCell[First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int ICell, int Kv) const {
      const int ME = M.MaxEdges;
      T Tmp = splat<T>(0.0);
      const int N = L.N[Le];
      for (int J = 0; J < N; ++J)
         Tmp -= L.DivC[Le * ME + J] * ldk<T>(Del2Edge, L.Edge[Le * ME + J], K, Kv);
      stk<T>(Del2Div, ICell, K, Kv, Tmp);
   }
};
void launchCellAuxState2(const MeshView &M, int K, const AuxPtrs &A, hipStream_t S) {
   Del2CellBody B{M, K, A.Del2Edge, A.Del2DivCell};
   launchTile(B, M.NCellsAll, K, S);
}

// ---------------------------------------------------------------------------------------
// cellAuxState3: LayerThicknessAuxVars::computeVarsOnCells (LayerThicknessAuxVars.h:63-82)
struct SshCellBody {
   MeshView M;
   int K;
   const Real *H;
   Real *Ssh;
   struct Lds {
      Real *BD;
   };
   size_t ldsBytes(int Tile) const { return ldsRound8(sizeof(Real) * Tile); }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      LdsCarver C{P};
      Lds L;
      L.BD = C.take<Real>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt; I += NThr)
         L.BD[I] = M.BottomDepth[First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int ICell, int Kv) const {
      stk<T>(Ssh, ICell, K, Kv, ldk<T>(H, ICell, K, Kv) - L.BD[Le]);
   }
};
void launchCellAuxState3(const MeshView &M, int K, const AuxPtrs &A, const Real *H, hipStream_t S) {
   SshCellBody B{M, K, H, A.SshCell};
   launchTile(B, M.NCellsAll, K, S);
}

// ---------------------------------------------------------------------------------------
// edgeAuxState4: TracerAuxVars::computeVarsOnEdge (auxiliaryVars/TracerAuxVars.h:25-59).
// Tracer loop inside the thread: h at the two cells (and u) are read once for all NT.
struct TracerEdgeBody {
   MeshView M;
   int K, NT;
   const Real *U, *H, *Tr;
   Real *HTr;
   int Upwind;
   struct Lds {
      int *C0, *C1;
   };
   size_t ldsBytes(int Tile) const { return ldsRound8(sizeof(int) * Tile) * 2; }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      LdsCarver C{P};
      Lds L;
      L.C0 = C.take<int>(Tile);
      L.C1 = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt; I += NThr) {
         L.C0[I] = M.CellsOnEdge[2 * (First + I)];
         L.C1[I] = M.CellsOnEdge[2 * (First + I) + 1];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IEdge, int Kv) const {
      const int C0 = L.C0[Le], C1 = L.C1[Le];
      const T H0 = ldk<T>(H, C0, K, Kv), H1 = ldk<T>(H, C1, K, Kv);
      T Ue = splat<T>(0.0);
      if (Upwind)
         Ue = ldk<T>(U, IEdge, K, Kv);
      const size_t TrStride = (size_t)M.NCellsSize * K, HTrStride = (size_t)M.NEdgesSize * K;
      for (int Lt = 0; Lt < NT; ++Lt) {
         const T HT0 = H0 * ldk<T>(Tr + Lt * TrStride, C0, K, Kv);
         const T HT1 = H1 * ldk<T>(Tr + Lt * TrStride, C1, K, Kv);
         T R;
         if (!Upwind)
            R = 0.5 * (HT0 + HT1);
         else
            R = upwind(Ue, HT0, HT1);
         stk<T>(HTr + Lt * HTrStride, IEdge, K, Kv, R);
      }
   }
};
void launchEdgeAuxState4(const MeshView &M, int K, int NT, const AuxPtrs &A, const Real *U, const Real *H,
                         const Real *Tr, int TracerUpwind, hipStream_t S) {
   if (NT <= 0)
      return;
   TracerEdgeBody B{M, K, NT, U, H, Tr, A.HTracersEdge, TracerUpwind};
   launchTile(B, M.NEdgesAll, K, S);
}

// ---------------------------------------------------------------------------------------
// cellAuxState4: TracerAuxVars::computeVarsOnCells (auxiliaryVars/TracerAuxVars.h:61-91)
struct TracerCellBody {
   MeshView M;
   int K, NT;
   const Real *HMean, *Tr;
   Real *Del2Tr;
   struct Lds {
      Real *Coef, *InvA;
      int *Edge, *C0, *C1, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME = M.MaxEdges;
      return ldsRound8(sizeof(Real) * Tile * ME) + ldsRound8(sizeof(Real) * Tile) +
             ldsRound8(sizeof(int) * Tile * ME) * 3 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *P, int Tile) const {
      const int ME = M.MaxEdges;
      LdsCarver C{P};
      Lds L;
      L.Coef = C.take<Real>(Tile * ME);
      L.InvA = C.take<Real>(Tile);
      L.Edge = C.take<int>(Tile * ME);
      L.C0   = C.take<int>(Tile * ME);
      L.C1   = C.take<int>(Tile * ME);
      L.N    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME = M.MaxEdges;
      for (int I = Tid; I < Cnt * ME; I += NThr) {
         const size_t G = (size_t)First * ME + I;
         L.Coef[I]      = M.Del2TrCoefOnCell[G];
         L.Edge[I]      = M.EdgesOnCell[G];
         L.C0[I]        = M.CellsOnEdgeOnCell[2 * G];
         L.C1[I]        = M.CellsOnEdgeOnCell[2 * G + 1];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         L.N[I]    = M.NEdgesOnCell[First + I];
         L.InvA[I] = M.InvAreaCell[First + I];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int ICell, int Kv) const {
      const int ME = M.MaxEdges;
      const int N  = L.N[Le];
      const size_t TrStride = (size_t)M.NCellsSize * K;
      for (int Lt = 0; Lt < NT; ++Lt) {
         const Real *TrL = Tr + Lt * TrStride;
         T Tmp           = splat<T>(0.0);
         for (int J = 0; J < N; ++J) {
            const T Grad = ldk<T>(TrL, L.C1[Le * ME + J], K, Kv) - ldk<T>(TrL, L.C0[Le * ME + J], K, Kv);
            Tmp -= L.Coef[Le * ME + J] * ldk<T>(HMean, L.Edge[Le * ME + J], K, Kv) * Grad;
         }
         stk<T>(Del2Tr + Lt * TrStride, ICell, K, Kv, Tmp * L.InvA[Le]);
      }
   }
};
void launchCellAuxState4(const MeshView &M, int K, int NT, const AuxPtrs &A, const Real *Tr, hipStream_t S) {
   if (NT <= 0)
      return;
   TracerCellBody B{M, K, NT, A.MeanLayerThickEdge, Tr, A.Del2TracersCell};
   launchTile(B, M.NCellsAll, K, S);
}

} // namespace OMEGA
