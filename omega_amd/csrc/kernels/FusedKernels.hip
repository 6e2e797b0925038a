// FusedKernels.hip -- Tendencies::computeAllTendencies as a fused RHS.
//
// The reference evaluates the RHS with 23 launches (components/omega/src/ocn/
// AuxiliaryState.cpp:79-182 + Tendencies.cpp:272-481), every intermediate going through
// HBM.  The data-dependency levels of the discretisation only force three
// global cuts, so the fused RHS is 6 kernels in 3 dependency levels:
//
//   L1  vertex: RelVort, NormRelVort, NormPlanetVort          (launchVertexAuxState1)
//       cell  : KE, VelocityDiv, LayerThicknessTend, Del2Tracers          (FusedCell1Body)
//   L2  cell  : Del2Div    vertex: Del2RelVort  (Del2Edge recomputed inline, never stored)
//   L3  edge  : NormalVelocityTend, all terms in registers, one store     (FusedEdgeBody)
//       cell  : TracerTend, all terms, tracer loop inside the thread      (FusedCell3Body)
//
// Edge-located intermediates of the reference (FluxLayerThickEdge, MeanLayerThickEdge,
// NormRelVortEdge, NormPlanetVortEdge, Del2Edge, HTracersEdge) and SshCell are recomputed
// where they are consumed, from the same inputs with the same operations in the same order,
// so every value -- and therefore every tendency -- is bit-identical to the unfused path.
// Compiled with -ffp-contract=off.
#include "KernelCommon.h"
#include "Kernels.h"

namespace OMEGA {

constexpr int MEMAX = 8; // register-array bound on edges per cell in the fused kernels

// ---------------------------------------------------------------------------------------
// L1 cell pass: KineticAuxVars::computeVarsOnCell (KineticAuxVars.h:20-47),
// LayerThicknessAuxVars::computeVarsOnEdge inline (LayerThicknessAuxVars.h:25-61) feeding
// ThicknessFluxDivOnCell (TendencyTerms.h:35-58), TracerAuxVars::computeVarsOnCells
// (TracerAuxVars.h:61-91).
struct FusedCell1Body {
   MeshView M;
   int K, NT;
   TendParams P;
   int DoDel2Tr;
   const Real *H, *U, *Tr;
   Real *KE, *Div, *HTend, *Del2Tr;
   struct Lds {
      Real *KEC, *DivC, *DvS, *D2T, *InvA;
      int *Edge, *C0, *C1, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME = M.MaxEdges;
      return ldsRound8(sizeof(Real) * Tile * ME) * 4 + ldsRound8(sizeof(Real) * Tile) +
             ldsRound8(sizeof(int) * Tile * ME) * 3 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      const int ME = M.MaxEdges;
      LdsCarver C{Ptr};
      Lds L;
      L.KEC  = C.take<Real>(Tile * ME);
      L.DivC = C.take<Real>(Tile * ME);
      L.DvS  = C.take<Real>(Tile * ME);
      L.D2T  = C.take<Real>(Tile * ME);
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
         L.KEC[I]       = M.KECoefOnCell[G];
         L.DivC[I]      = M.DivCoefOnCell[G];
         L.DvS[I]       = M.DvSignOnCell[G];
         L.D2T[I]       = M.Del2TrCoefOnCell[G];
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
      const int ME    = M.MaxEdges;
      const int N     = L.N[Le];
      const Real InvA = L.InvA[Le];
      T KETmp = splat<T>(0.0), DivTmp = splat<T>(0.0), HDivTmp = splat<T>(0.0);
      T HMeanJ[MEMAX];
#pragma unroll
      for (int J = 0; J < MEMAX; ++J) {
         HMeanJ[J] = splat<T>(0.0);
         if (J < N) {
            const int JEdge = L.Edge[Le * ME + J];
            const T Ue      = ldk<T>(U, JEdge, K, Kv);
            const T H0 = ldk<T>(H, L.C0[Le * ME + J], K, Kv), H1 = ldk<T>(H, L.C1[Le * ME + J], K, Kv);
            const T Mean = 0.5 * (H0 + H1);
            HMeanJ[J]    = Mean;
            const T Flux = P.FluxThicknessUpwind ? upwind(Ue, H0, H1) : Mean;
            KETmp += L.KEC[Le * ME + J] * Ue * Ue;
            DivTmp -= L.DivC[Le * ME + J] * Ue;
            HDivTmp -= L.DvS[Le * ME + J] * Flux * Ue * InvA;
         }
      }
      stk<T>(KE, ICell, K, Kv, KETmp);
      stk<T>(Div, ICell, K, Kv, DivTmp);
      T HT = splat<T>(0.0);
      if (P.ThicknessFluxTendencyEnable)
         HT -= HDivTmp;
      stk<T>(HTend, ICell, K, Kv, HT);
      if (DoDel2Tr) {
         const size_t CStride = (size_t)M.NCellsSize * K;
         for (int Lt = 0; Lt < NT; ++Lt) {
            const Real *TrL = Tr + Lt * CStride;
            T Tmp           = splat<T>(0.0);
#pragma unroll
            for (int J = 0; J < MEMAX; ++J) {
               if (J < N) {
                  const T Grad = ldk<T>(TrL, L.C1[Le * ME + J], K, Kv) - ldk<T>(TrL, L.C0[Le * ME + J], K, Kv);
                  Tmp -= L.D2T[Le * ME + J] * HMeanJ[J] * Grad;
               }
            }
            stk<T>(Del2Tr + Lt * CStride, ICell, K, Kv, Tmp * InvA);
         }
      }
   }
};

// ---------------------------------------------------------------------------------------
// L2 cell pass: VelocityDel2AuxVars::computeVarsOnCell (VelocityDel2AuxVars.h:47-67) with
// Del2Edge (computeVarsOnEdge, :21-45) evaluated inline at each edge of the cell.
struct FusedDel2CellBody {
   MeshView M;
   int K;
   const Real *Div, *RelVort;
   Real *Del2Div;
   struct Lds {
      Real *DivC, *InvDc, *InvDv2, *Mask;
      int *C0, *C1, *V0, *V1, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME = M.MaxEdges;
      return ldsRound8(sizeof(Real) * Tile * ME) * 4 + ldsRound8(sizeof(int) * Tile * ME) * 4 +
             ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      const int ME = M.MaxEdges;
      LdsCarver C{Ptr};
      Lds L;
      L.DivC   = C.take<Real>(Tile * ME);
      L.InvDc  = C.take<Real>(Tile * ME);
      L.InvDv2 = C.take<Real>(Tile * ME);
      L.Mask   = C.take<Real>(Tile * ME);
      L.C0     = C.take<int>(Tile * ME);
      L.C1     = C.take<int>(Tile * ME);
      L.V0     = C.take<int>(Tile * ME);
      L.V1     = C.take<int>(Tile * ME);
      L.N      = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME = M.MaxEdges;
      for (int I = Tid; I < Cnt * ME; I += NThr) {
         const size_t G = (size_t)First * ME + I;
         const int E    = M.EdgesOnCell[G];
         L.DivC[I]      = M.DivCoefOnCell[G];
         L.InvDc[I]     = M.InvDcEdge[E];
         L.InvDv2[I]    = M.InvDvEdgeDel2[E];
         L.Mask[I]      = M.EdgeMask1D[E];
         L.C0[I]        = M.CellsOnEdge[2 * E];
         L.C1[I]        = M.CellsOnEdge[2 * E + 1];
         L.V0[I]        = M.VerticesOnEdge[2 * E];
         L.V1[I]        = M.VerticesOnEdge[2 * E + 1];
      }
      for (int I = Tid; I < Cnt; I += NThr)
         L.N[I] = M.NEdgesOnCell[First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int ICell, int Kv) const {
      const int ME = M.MaxEdges;
      const int N  = L.N[Le];
      T Tmp        = splat<T>(0.0);
      for (int J = 0; J < N; ++J) {
         const int I     = Le * ME + J;
         const T GradDiv = (ldk<T>(Div, L.C1[I], K, Kv) - ldk<T>(Div, L.C0[I], K, Kv)) * L.InvDc[I];
         const T CurlVort = -(ldk<T>(RelVort, L.V1[I], K, Kv) - ldk<T>(RelVort, L.V0[I], K, Kv)) * L.InvDv2[I];
         const T Del2E   = L.Mask[I] * GradDiv + CurlVort;
         Tmp -= L.DivC[I] * Del2E;
      }
      stk<T>(Del2Div, ICell, K, Kv, Tmp);
   }
};

// L2 vertex pass: VelocityDel2AuxVars::computeVarsOnVertex (VelocityDel2AuxVars.h:69-89)
struct FusedDel2VertexBody {
   MeshView M;
   int K;
   const Real *Div, *RelVort;
   Real *Del2RelVort;
   struct Lds {
      Real *VortC, *InvDc, *InvDv2, *Mask;
      int *C0, *C1, *V0, *V1;
   };
   size_t ldsBytes(int Tile) const {
      const int VD = M.VertexDegree;
      return ldsRound8(sizeof(Real) * Tile * VD) * 4 + ldsRound8(sizeof(int) * Tile * VD) * 4;
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      const int VD = M.VertexDegree;
      LdsCarver C{Ptr};
      Lds L;
      L.VortC  = C.take<Real>(Tile * VD);
      L.InvDc  = C.take<Real>(Tile * VD);
      L.InvDv2 = C.take<Real>(Tile * VD);
      L.Mask   = C.take<Real>(Tile * VD);
      L.C0     = C.take<int>(Tile * VD);
      L.C1     = C.take<int>(Tile * VD);
      L.V0     = C.take<int>(Tile * VD);
      L.V1     = C.take<int>(Tile * VD);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int VD = M.VertexDegree;
      for (int I = Tid; I < Cnt * VD; I += NThr) {
         const size_t G = (size_t)First * VD + I;
         const int E    = M.EdgesOnVertex[G];
         L.VortC[I]     = M.VortCoefOnVertex[G];
         L.InvDc[I]     = M.InvDcEdge[E];
         L.InvDv2[I]    = M.InvDvEdgeDel2[E];
         L.Mask[I]      = M.EdgeMask1D[E];
         L.C0[I]        = M.CellsOnEdge[2 * E];
         L.C1[I]        = M.CellsOnEdge[2 * E + 1];
         L.V0[I]        = M.VerticesOnEdge[2 * E];
         L.V1[I]        = M.VerticesOnEdge[2 * E + 1];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IVertex, int Kv) const {
      const int VD = M.VertexDegree;
      T Tmp        = splat<T>(0.0);
      for (int J = 0; J < VD; ++J) {
         const int I     = Le * VD + J;
         const T GradDiv = (ldk<T>(Div, L.C1[I], K, Kv) - ldk<T>(Div, L.C0[I], K, Kv)) * L.InvDc[I];
         const T CurlVort = -(ldk<T>(RelVort, L.V1[I], K, Kv) - ldk<T>(RelVort, L.V0[I], K, Kv)) * L.InvDv2[I];
         const T Del2E   = L.Mask[I] * GradDiv + CurlVort;
         Tmp += L.VortC[I] * Del2E;
      }
      stk<T>(Del2RelVort, IVertex, K, Kv, Tmp);
   }
};

// ---------------------------------------------------------------------------------------
// L3 edge pass: every velocity term (TendencyTerms.h:81-334) in registers.  The edge-located
// inputs of PotentialVortHAdvOnEdge at each EdgesOnEdge neighbour (FluxLayerThickEdge,
// NormRelVortEdge, NormPlanetVortEdge) are rebuilt from h at its two cells and the
// normalised vorticities at its two vertices (LayerThicknessAuxVars.h:25-61,
// VorticityAuxVars.h:61-76); SshCell from h - BottomDepth (LayerThicknessAuxVars.h:63-82).
struct FusedEdgeBody {
   MeshView M;
   int K;
   TendParams P;
   const Real *H, *U;
   const Real *RelVort, *NormRelVortV, *NormPlanetVortV, *KE, *Div, *Del2Div, *Del2RelVort, *NormalStress;
   Real *Tend;
   struct Lds {
      Real *W, *InvDc, *InvDv, *Mask, *MaskGrav, *C2, *C4, *BD0, *BD1;
      int *EoE, *PVS, *C0, *C1, *V0, *V1, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME2 = M.MaxEdges2;
      return ldsRound8(sizeof(Real) * Tile * ME2) + ldsRound8(sizeof(Real) * Tile) * 8 +
             ldsRound8(sizeof(int) * Tile * ME2) + ldsRound8(sizeof(int) * Tile * ME2 * 4) +
             ldsRound8(sizeof(int) * Tile) * 5;
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      const int ME2 = M.MaxEdges2;
      LdsCarver C{Ptr};
      Lds L;
      L.W        = C.take<Real>(Tile * ME2);
      L.InvDc    = C.take<Real>(Tile);
      L.InvDv    = C.take<Real>(Tile);
      L.Mask     = C.take<Real>(Tile);
      L.MaskGrav = C.take<Real>(Tile);
      L.C2       = C.take<Real>(Tile);
      L.C4       = C.take<Real>(Tile);
      L.BD0      = C.take<Real>(Tile);
      L.BD1      = C.take<Real>(Tile);
      L.EoE      = C.take<int>(Tile * ME2);
      L.PVS      = C.take<int>(Tile * ME2 * 4);
      L.C0       = C.take<int>(Tile);
      L.C1       = C.take<int>(Tile);
      L.V0       = C.take<int>(Tile);
      L.V1       = C.take<int>(Tile);
      L.N        = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const int ME2   = M.MaxEdges2;
      const Real Grav = 9.80665; // TendencyTerms.h:176
      for (int I = Tid; I < Cnt * ME2; I += NThr) {
         const size_t G = (size_t)First * ME2 + I;
         L.W[I]         = M.WeightsOnEdge[G];
         L.EoE[I]       = M.EdgesOnEdge[G];
      }
      for (int I = Tid; I < Cnt * ME2 * 4; I += NThr)
         L.PVS[I] = M.PVStencil[(size_t)First * ME2 * 4 + I];
      for (int I = Tid; I < Cnt; I += NThr) {
         const int E     = First + I;
         const Real Mask = M.EdgeMask1D[E];
         const int C0 = M.CellsOnEdge[2 * E], C1 = M.CellsOnEdge[2 * E + 1];
         L.InvDc[I]    = M.InvDcEdge[E];
         L.InvDv[I]    = M.InvDvEdge[E];
         L.Mask[I]     = Mask;
         L.MaskGrav[I] = Mask * Grav;
         L.C2[I]       = Mask * P.ViscDel2 * M.MeshScalingDel2[E];
         L.C4[I]       = Mask * P.ViscDel4 * M.MeshScalingDel4[E];
         L.BD0[I]      = M.BottomDepth[C0];
         L.BD1[I]      = M.BottomDepth[C1];
         L.C0[I]       = C0;
         L.C1[I]       = C1;
         L.V0[I]       = M.VerticesOnEdge[2 * E];
         L.V1[I]       = M.VerticesOnEdge[2 * E + 1];
         L.N[I]        = M.NEdgesOnEdge[E];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IEdge, int Kv) const {
      const int ME2 = M.MaxEdges2;
      const int C0 = L.C0[Le], C1 = L.C1[Le], V0 = L.V0[Le], V1 = L.V1[Le];
      const Real InvDc = L.InvDc[Le], InvDv = L.InvDv[Le];
      const T H0 = ldk<T>(H, C0, K, Kv), H1 = ldk<T>(H, C1, K, Kv);
      T TendV = splat<T>(0.0);
      if (P.PVTendencyEnable) {
         // NormRelVortEdge / NormPlanetVortEdge of this edge (VorticityAuxVars.h:68-74)
         const T QRe = 0.5 * (ldk<T>(NormRelVortV, V0, K, Kv) + ldk<T>(NormRelVortV, V1, K, Kv));
         const T QFe = 0.5 * (ldk<T>(NormPlanetVortV, V0, K, Kv) + ldk<T>(NormPlanetVortV, V1, K, Kv));
         T VortTmp   = splat<T>(0.0);
         const int N = L.N[Le];
         for (int J = 0; J < N; ++J) {
            const int I     = Le * ME2 + J;
            const int JEdge = L.EoE[I];
            const int *S4   = &L.PVS[I * 4];
            const T Uj      = ldk<T>(U, JEdge, K, Kv);
            const T Hj0 = ldk<T>(H, S4[0], K, Kv), Hj1 = ldk<T>(H, S4[1], K, Kv);
            const T Flux = P.FluxThicknessUpwind ? upwind(Uj, Hj0, Hj1) : T(0.5 * (Hj0 + Hj1));
            const T QRj  = 0.5 * (ldk<T>(NormRelVortV, S4[2], K, Kv) + ldk<T>(NormRelVortV, S4[3], K, Kv));
            const T QFj  = 0.5 * (ldk<T>(NormPlanetVortV, S4[2], K, Kv) + ldk<T>(NormPlanetVortV, S4[3], K, Kv));
            const T NormVort = (QRe + QFe + QRj + QFj) * 0.5;
            VortTmp += L.W[I] * Flux * Uj * NormVort;
         }
         TendV += L.Mask[Le] * VortTmp;
      }
      if (P.KETendencyEnable)
         TendV -= L.Mask[Le] * (ldk<T>(KE, C1, K, Kv) - ldk<T>(KE, C0, K, Kv)) * InvDc;
      if (P.SSHTendencyEnable) {
         const T Ssh0 = H0 - L.BD0[Le], Ssh1 = H1 - L.BD1[Le];
         TendV -= L.MaskGrav[Le] * (Ssh1 - Ssh0) * InvDc;
      }
      if (P.VelDiffTendencyEnable) {
         const T Del2U = ((ldk<T>(Div, C1, K, Kv) - ldk<T>(Div, C0, K, Kv)) * InvDc -
                          (ldk<T>(RelVort, V1, K, Kv) - ldk<T>(RelVort, V0, K, Kv)) * InvDv);
         TendV += L.C2[Le] * Del2U;
      }
      if (P.VelHyperDiffTendencyEnable) {
         const T Del2U = (P.DivFactor * (ldk<T>(Del2Div, C1, K, Kv) - ldk<T>(Del2Div, C0, K, Kv)) * InvDc -
                          (ldk<T>(Del2RelVort, V1, K, Kv) - ldk<T>(Del2RelVort, V0, K, Kv)) * InvDv);
         TendV -= L.C4[Le] * Del2U;
      }
      constexpr int W = VecW<T>::W;
      if (P.WindForcingTendencyEnable && Kv == 0) {
         const Real HMean0       = 0.5 * (getc(H0, 0) + getc(H1, 0));
         const Real InvThickEdge = 1. / HMean0;
         setc(TendV, 0, getc(TendV, 0) + L.Mask[Le] * InvThickEdge * NormalStress[IEdge] / P.Density0);
      }
      if (P.BottomDragTendencyEnable && (Kv + 1) * W >= K) {
         const int KBot          = K - 1;
         const int Comp          = KBot - Kv * W;
         const Real VelNormEdge  = sqrt(KE[(size_t)C0 * K + KBot] + KE[(size_t)C1 * K + KBot]);
         const Real HMeanB       = 0.5 * (getc(H0, Comp) + getc(H1, Comp));
         const Real InvThickEdge = 1. / HMeanB;
         setc(TendV, Comp,
              getc(TendV, Comp) - L.Mask[Le] * P.BottomDragCoeff * VelNormEdge * InvThickEdge * U[(size_t)IEdge * K + KBot]);
      }
      stk<T>(Tend, IEdge, K, Kv, TendV);
   }
};

// ---------------------------------------------------------------------------------------
// L3 cell pass: tracer tendencies (TendencyTerms.h:349-480) with HTracersEdge
// (TracerAuxVars.h:25-59) and MeanLayerThickEdge rebuilt inline; tracer loop inside.
struct FusedCell3Body {
   MeshView M;
   int K, NT;
   TendParams P;
   const Real *H, *U, *Tr, *Del2Tr;
   Real *Tend;
   struct Lds {
      Real *MDvS, *Df2, *Df4, *InvA;
      int *Edge, *C0, *C1, *N;
   };
   size_t ldsBytes(int Tile) const {
      const int ME = M.MaxEdges;
      return ldsRound8(sizeof(Real) * Tile * ME) * 3 + ldsRound8(sizeof(Real) * Tile) +
             ldsRound8(sizeof(int) * Tile * ME) * 3 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      const int ME = M.MaxEdges;
      LdsCarver C{Ptr};
      Lds L;
      L.MDvS = C.take<Real>(Tile * ME);
      L.Df2  = C.take<Real>(Tile * ME);
      L.Df4  = C.take<Real>(Tile * ME);
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
         L.MDvS[I]      = M.MaskDvSignOnCell[G];
         L.Df2[I]       = M.Diff2CoefOnCell[G];
         L.Df4[I]       = M.Diff4CoefOnCell[G];
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
      const int ME    = M.MaxEdges;
      const int N     = L.N[Le];
      const Real InvA = L.InvA[Le];
      T UJ[MEMAX], H0J[MEMAX], H1J[MEMAX];
#pragma unroll
      for (int J = 0; J < MEMAX; ++J) {
         UJ[J] = H0J[J] = H1J[J] = splat<T>(0.0);
         if (J < N) {
            UJ[J]  = ldk<T>(U, L.Edge[Le * ME + J], K, Kv);
            H0J[J] = ldk<T>(H, L.C0[Le * ME + J], K, Kv);
            H1J[J] = ldk<T>(H, L.C1[Le * ME + J], K, Kv);
         }
      }
      const size_t CStride = (size_t)M.NCellsSize * K;
      for (int Lt = 0; Lt < NT; ++Lt) {
         const Real *TrL = Tr + Lt * CStride;
         const Real *D2L = Del2Tr + Lt * CStride;
         T HAdvTmp = splat<T>(0.0), DiffTmp = splat<T>(0.0), HypTmp = splat<T>(0.0);
#pragma unroll
         for (int J = 0; J < MEMAX; ++J) {
            if (J < N) {
               const int I   = Le * ME + J;
               const int JC0 = L.C0[I], JC1 = L.C1[I];
               const T T0 = ldk<T>(TrL, JC0, K, Kv), T1 = ldk<T>(TrL, JC1, K, Kv);
               if (P.TracerHorzAdvTendencyEnable) {
                  const T HT0 = H0J[J] * T0, HT1 = H1J[J] * T1;
                  const T HTr = P.FluxTracerUpwind ? upwind(UJ[J], HT0, HT1) : T(0.5 * (HT0 + HT1));
                  HAdvTmp -= L.MDvS[I] * HTr * UJ[J] * InvA;
               }
               if (P.TracerDiffTendencyEnable) {
                  const T Mean = 0.5 * (H0J[J] + H1J[J]);
                  DiffTmp -= L.Df2[I] * Mean * (T1 - T0);
               }
               if (P.TracerHyperDiffTendencyEnable)
                  HypTmp -= L.Df4[I] * (ldk<T>(D2L, JC1, K, Kv) - ldk<T>(D2L, JC0, K, Kv));
            }
         }
         T TendV = splat<T>(0.0);
         if (P.TracerHorzAdvTendencyEnable)
            TendV -= HAdvTmp;
         if (P.TracerDiffTendencyEnable)
            TendV += P.EddyDiff2 * DiffTmp * InvA;
         if (P.TracerHyperDiffTendencyEnable)
            TendV -= P.EddyDiff4 * HypTmp * InvA;
         stk<T>(Tend + Lt * CStride, ICell, K, Kv, TendV);
      }
   }
};

// ---------------------------------------------------------------------------------------
void launchFusedRHS(const MeshView &M, int K, int NT, const TendParams &P, const AuxPtrs &A, Real *HTend, Real *UTend,
                    Real *TrTend, const Real *H, const Real *U, const Real *Tr, hipStream_t S) {
   // L1
   launchVertexAuxState1(M, K, A, H, U, S);
   const int DoDel2Tr = (NT > 0 && P.TracerHyperDiffTendencyEnable) ? 1 : 0;
   {
      FusedCell1Body B{M, K, NT, P, DoDel2Tr, H, U, Tr, A.KineticEnergyCell, A.VelocityDivCell, HTend, A.Del2TracersCell};
      launchTile(B, M.NCellsAll, K, S);
   }
   if (P.WindForcingTendencyEnable)
      launchEdgeAuxState1(M, A, P.WindInterpIsotropic, S);
   // L2 (only the del4 term consumes it)
   if (P.VelHyperDiffTendencyEnable) {
      FusedDel2CellBody BC{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2DivCell};
      launchTile(BC, M.NCellsAll, K, S);
      FusedDel2VertexBody BV{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2RelVortVertex};
      launchTile(BV, M.NVerticesAll, K, S);
   }
   // L3
   {
      FusedEdgeBody B{M,       K,           P,           H,           U,
                      A.RelVortVertex, A.NormRelVortVertex, A.NormPlanetVortVertex, A.KineticEnergyCell, A.VelocityDivCell,
                      A.Del2DivCell,   A.Del2RelVortVertex, A.NormalStressEdge,     UTend};
      launchTile(B, M.NEdgesAll, K, S);
   }
   if (NT > 0) {
      FusedCell3Body B{M, K, NT, P, H, U, Tr, A.Del2TracersCell, TrTend};
      launchTile(B, M.NCellsAll, K, S);
   }
}

} // namespace OMEGA
