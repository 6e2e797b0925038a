// FusedKernelsImpl.h -- the bodies of the fused right-hand side and the template that launches them (launchFusedT).
// Included by FusedKernels.hip (the dispatcher, which only DECLARES the instantiations it calls: `extern template`) and by
// the FusedInst*.hip translation units, each of which instantiates a few of them -- so that the 17 instantiations of a
// 600-line launcher over ~2000 lines of kernel bodies compile in parallel (as one translation unit: 4-5 minutes).
#ifndef OMEGA_AMD_FUSEDKERNELSIMPL_H
#define OMEGA_AMD_FUSEDKERNELSIMPL_H
// Tendencies::computeAllTendencies as a fused RHS (the text below describes the dependency levels; DESIGN.md section 4 the current bodies).
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
#include "../Pacer.h"

#include <cstdlib>
#include <functional>
#include <type_traits>

// tuning knobs of the tracer cell kernels (VGPR budget / levels per thread)
#ifndef OMEGA_CELL_MINW
#define OMEGA_CELL_MINW 2
#endif
#ifndef OMEGA_CELL_MAXW
#define OMEGA_CELL_MAXW 2
#endif

#ifndef OMEGA_PVF_MINW
#define OMEGA_PVF_MINW OMEGA_CELL_MINW
#endif
#ifndef OMEGA_C3_MINW
#define OMEGA_C3_MINW OMEGA_CELL_MINW
#endif
#ifndef OMEGA_EDGE_MINW
#define OMEGA_EDGE_MINW 2
#endif
#ifndef OMEGA_EDGE_MAXW
#define OMEGA_EDGE_MAXW 2
#endif

namespace OMEGA {

// ---------------------------------------------------------------------------------------
// Addressing.  Inside one array plane ([rows][pitch] doubles) an element is addressed by a 32-bit BYTE offset from
// the plane's base pointer, and every access is a buffer (MUBUF) instruction: the wave-uniform base (kernel argument,
// or argument + tracer * plane size) goes into a buffer resource -- 4 SGPRs, no stride, no bound below 4 GiB - 256 B
// (fusedRHSSupported checks the planes are smaller) -- and the offset is the instruction's VGPR offset:
// `buffer_load_dwordx4 v, v_off, s[rsrc], 0 offen`.  A gather costs one 32-bit VGPR per neighbour, shared by every
// array of that index space (h, each tracer, each Del2Tracers plane, ...), and no address arithmetic.  (Written as
// pointer arithmetic, two thirds of the accesses became 64-bit VGPR address computations and the tracer planes
// flat_load: DESIGN.md section 4.)
// Every resource has the size BufOOB: offsets below it are in range for any plane; a lane whose offset IS BufOOB is
// out of range -- its load returns 0 without touching memory (checked on the hardware).  ldoIf uses that to switch a
// load off by a (wave-uniform or per-lane) condition without a branch, so it can be issued early with the others.
constexpr unsigned BufOOB = FusedMaxPlaneBytes;
typedef unsigned BufV4 __attribute__((ext_vector_type(4)));
typedef unsigned BufV2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t bufRsrc(const Real *Base) {
   return __builtin_amdgcn_make_buffer_rsrc(const_cast<Real *>(Base), 0, BufOOB, 0x00020000);
}
template <class T> __device__ __forceinline__ T ldo(const Real *Base, unsigned ByteOff) {
   if constexpr (sizeof(T) == 16)
      return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(bufRsrc(Base), ByteOff, 0, 0));
   else
      return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(bufRsrc(Base), ByteOff, 0, 0));
}
template <class T> __device__ __forceinline__ void sto(Real *Base, unsigned ByteOff, T V) {
   if constexpr (sizeof(T) == 16)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(BufV4, V), bufRsrc(Base), ByteOff, 0, 0);
   else
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(BufV2, V), bufRsrc(Base), ByteOff, 0, 0);
}
/// streaming store (nt) for outputs nobody re-reads inside the same kernel
template <class T> __device__ __forceinline__ void stnt(Real *Base, unsigned ByteOff, T V) {
   if constexpr (sizeof(T) == 16)
      __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(BufV4, V), bufRsrc(Base), ByteOff, 0, 2);
   else
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(BufV2, V), bufRsrc(Base), ByteOff, 0, 2);
}
template <class T> __device__ __forceinline__ void stoIf(bool Cond, Real *Base, unsigned ByteOff, T V) {
   sto<T>(Base, Cond ? ByteOff : BufOOB, V); // (an out-of-range store is dropped)
}
template <class T> __device__ __forceinline__ void stntIf(bool Cond, Real *Base, unsigned ByteOff, T V) {
   stnt<T>(Base, Cond ? ByteOff : BufOOB, V);
}
template <class T> __device__ __forceinline__ T ldoIf(bool Cond, const Real *Base, unsigned ByteOff) {
   return ldo<T>(Base, Cond ? ByteOff : BufOOB);
}
/// streaming load (nt) for values this launch reads exactly once (running PV sums, stage-update operands)
template <class T> __device__ __forceinline__ T ldnt(const Real *Base, unsigned ByteOff) {
   if constexpr (sizeof(T) == 16)
      return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b128(bufRsrc(Base), ByteOff, 0, 2));
   else
      return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b64(bufRsrc(Base), ByteOff, 0, 2));
}
template <class T> __device__ __forceinline__ T ldntIf(bool Cond, const Real *Base, unsigned ByteOff) {
   return ldnt<T>(Base, Cond ? ByteOff : BufOOB);
}
/// 16 bytes per lane from a buffer straight into LDS (`buffer_load_dwordx4 ... lds`, gfx950): lane l of the wavefront
/// lands at LdsWaveBase + 16 l; no VGPR is written, the load counts on vmcnt like any other; an out-of-range offset
/// (BufOOB) moves nothing.  LdsWaveBase must be wave-uniform.
__device__ __forceinline__ void ldsDma16(const Real *Base, unsigned ByteOff, unsigned char *LdsWaveBase) {
   __builtin_amdgcn_raw_ptr_buffer_load_lds(bufRsrc(Base), (__attribute__((address_space(3))) void *)LdsWaveBase, 16, ByteOff,
                                            0, 0, 0);
}
/// keeps loop-invariant LDS reads inside the loop (a register each otherwise): nothing after this point may be
/// assumed unchanged in memory
__device__ __forceinline__ void loopFence() { __asm__ volatile("" ::: "memory"); }
/// Device-side view of one variable's Runge-Kutta stage update (Kernels.h: StageUpdate)
struct StageEpi {
   Real CB = 0, CA = 0;
   int First = 0, Last = 0, StoreTend = 0;
   Real *Next      = nullptr;
   const Real *Cur = nullptr;
   Real *Prov      = nullptr;
   const Real *CurH = nullptr, *ProvH = nullptr, *NextH = nullptr; // tracer kernel only
};
/// h / u: Next = (First ? Cur : Next) + CB*Tend; Prov = Cur + CA*Tend.  `CurIfFirst` is the RHS input at
/// this element, which IS the current state in the first stage (HaveReg: the caller holds it in registers).
template <class T, bool HaveReg>
__device__ __forceinline__ void stageUpdate(const StageEpi &E, unsigned Off, T Tend, T CurIfFirst) {
   const bool UseReg = HaveReg && E.First;
   T Base;
   if (E.First)
      Base = UseReg ? CurIfFirst : ldo<T>(E.Cur, Off);
   else
      Base = ldo<T>(E.Next, Off);
   stnt<T>(E.Next, Off, Base + E.CB * Tend);
   if (!E.Last) {
      const T Cur = E.First ? Base : ldo<T>(E.Cur, Off);
      stnt<T>(E.Prov, Off, Cur + E.CA * Tend);
   }
}
/// The same update with its loads split off, so that they can be asked for together with the loads of the
/// arithmetic that produces Tend (stagePre) instead of after it; a load the stage does not need is switched off
/// through its offset (ldoIf).  Same expressions as stageUpdate.
template <class T> struct StagePre {
   T NextOld, CurV;
};
template <class T, bool HaveReg> __device__ __forceinline__ StagePre<T> stagePre(const StageEpi &E, unsigned Off) {
   StagePre<T> R;
   R.NextOld = ldntIf<T>(!E.First, E.Next, Off);
   R.CurV    = ldntIf<T>(E.First ? !HaveReg : !E.Last, E.Cur, Off);
   return R;
}
template <class T, bool HaveReg>
__device__ __forceinline__ void stageApply(const StageEpi &E, unsigned Off, T Tend, T CurIfFirst, const StagePre<T> &R) {
   const T Base = E.First ? (HaveReg ? CurIfFirst : R.CurV) : R.NextOld;
   stnt<T>(E.Next, Off, Base + E.CB * Tend);
   if (!E.Last) {
      const T Cur = E.First ? Base : R.CurV;
      stnt<T>(E.Prov, Off, Cur + E.CA * Tend);
   }
}
template <class T> __device__ __forceinline__ unsigned rowOff(int Row, int K, int Kv) {
   return ((unsigned)Row * (unsigned)K + (unsigned)Kv * (unsigned)VecW<T>::W) * 8u;
}
/// make a wave-uniform pointer provably scalar (SGPR pair) so gathers use the
/// `saddr + 32-bit voffset` form instead of per-lane 64-bit pointers
__device__ __forceinline__ const Real *uniformPtr(const Real *P) {
   const unsigned long long V = reinterpret_cast<unsigned long long>(P);
   const unsigned Lo = __builtin_amdgcn_readfirstlane((unsigned)V);
   const unsigned Hi = __builtin_amdgcn_readfirstlane((unsigned)(V >> 32));
   return reinterpret_cast<const Real *>(((unsigned long long)Hi << 32) | Lo);
}
__device__ __forceinline__ Real *uniformPtr(Real *P) {
   return const_cast<Real *>(uniformPtr(const_cast<const Real *>(P)));
}
__device__ __forceinline__ double pick(bool C, double A, double B) { return C ? A : B; }
__device__ __forceinline__ dv2 pick(bool C, dv2 A, dv2 B) { return C ? A : B; }

// Cell kernels work on the TME (= MaxEdges, compile time) edge slots of a cell; slots past
// NEdgesOnCell carry zero coefficients and point at the zero sentinel rows, so the sweep is
// branch-free, all gathers of a sweep are in flight together, and padded slots add exact zeros.
// For each slot the cell across the edge and whether this cell is the edge's first cell come
// packed from NbrFlagOnCell: the value at "this" cell is loaded once and the (cell0, cell1)
// pair the reference indexes is rebuilt with a select.
//
// `Fast` = Default.yml term set (every term on, center fluxes, no wind / drag): the option
// flags fold at compile time and the loops carry no branches; otherwise they are read at run
// time (same arithmetic, more control flow).

// ---------------------------------------------------------------------------------------
// L1 cell pass: KineticAuxVars::computeVarsOnCell (KineticAuxVars.h:20-47),
// LayerThicknessAuxVars::computeVarsOnEdge inline (LayerThicknessAuxVars.h:25-61) feeding
// ThicknessFluxDivOnCell (TendencyTerms.h:35-58), TracerAuxVars::computeVarsOnCells
// (TracerAuxVars.h:61-91).
template <int TME, bool Fast, bool EPI = false> struct FusedCell1Body {
   static constexpr int MinWaves = OMEGA_CELL_MINW;
   static constexpr int MaxW     = OMEGA_CELL_MAXW;
   MeshView M;
   int K, NT;
   TendParams P;
   int DoDel2Tr;
   const Real *H, *U, *Tr;
   Real *KE, *Div, *HTend, *Del2Tr;
   StageEpi E{}; // thickness stage update (EPI)
   const int *List = nullptr; // optional cell list (the wide cells of a mesh with narrow tables: launchFusedT)
   struct Lds {
      Real *KEC, *DivC, *DvS, *D2T, *InvA;
      int *Edge, *NbrF, *N;
   };
   size_t ldsBytes(int Tile) const {
      return ldsRound8(sizeof(Real) * Tile * TME) * 4 + ldsRound8(sizeof(Real) * Tile) +
             ldsRound8(sizeof(int) * Tile * TME) * 2 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.KEC  = C.take<Real>(Tile * TME);
      L.DivC = C.take<Real>(Tile * TME);
      L.DvS  = C.take<Real>(Tile * TME);
      L.D2T  = C.take<Real>(Tile * TME);
      L.InvA = C.take<Real>(Tile);
      L.Edge = C.take<int>(Tile * TME);
      L.NbrF = C.take<int>(Tile * TME);
      L.N    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt * TME; I += NThr) {
         const int Cl   = I / TME;
         const int C    = List ? List[First + Cl] : First + Cl;
         const size_t G = (size_t)C * TME + (I - Cl * TME);
         L.KEC[I]       = M.KECoefOnCell[G];
         L.DivC[I]      = M.DivCoefOnCell[G];
         L.DvS[I]       = M.DvSignOnCell[G];
         L.D2T[I]       = Fast ? M.Del2TrCoefSOnCell[G] : M.Del2TrCoefOnCell[G];
         L.Edge[I]      = M.EdgesOnCell[G];
         L.NbrF[I]      = M.NbrFlagOnCell[G];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         const int C = List ? List[First + I] : First + I;
         L.InvA[I]   = M.InvAreaCell[C];
         L.N[I]      = M.NEdgesOnCell[C];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      if (L.N[Le] > TME)
         return; // a cell wider than these tables: it has its own (list) launch on the wide tables
      const int ICell       = List ? List[IElem] : IElem;
      const bool FluxUpwind = Fast ? false : (P.FluxThicknessUpwind != 0);
      const bool ThickOn    = Fast ? true : (P.ThicknessFluxTendencyEnable != 0);
      const Real InvA       = L.InvA[Le];
      const unsigned OffS   = rowOff<T>(ICell, K, Kv);
      unsigned OffN[TME];
      bool IsC0[TME];
      T Ue[TME], Hn[TME];
#pragma unroll
      for (int J = 0; J < TME; ++J) {
         const int F = L.NbrF[Le * TME + J];
         OffN[J]     = rowOff<T>(F & 0x3fffffff, K, Kv);
         IsC0[J]     = (F >> 30) != 0;
         Ue[J]       = ldo<T>(U, rowOff<T>(L.Edge[Le * TME + J], K, Kv));
         Hn[J]       = ldo<T>(H, OffN[J]);
      }
      const T Hs = ldo<T>(H, OffS);
      T KETmp = splat<T>(0.0), DivTmp = splat<T>(0.0), HDivTmp = splat<T>(0.0);
      T HMeanJ[TME];
#pragma unroll
      for (int J = 0; J < TME; ++J) {
         const T Mean = 0.5 * (Hs + Hn[J]); // 0.5*(h(c0)+h(c1)): a+b == b+a
         HMeanJ[J]    = Mean;
         T Flux       = Mean;
         if (FluxUpwind)
            Flux = upwind(Ue[J], pick(IsC0[J], Hs, Hn[J]), pick(IsC0[J], Hn[J], Hs));
         KETmp += L.KEC[Le * TME + J] * Ue[J] * Ue[J];
         DivTmp -= L.DivC[Le * TME + J] * Ue[J];
         HDivTmp -= L.DvS[Le * TME + J] * Flux * Ue[J] * InvA;
      }
      stnt<T>(KE, OffS, KETmp);
      stnt<T>(Div, OffS, DivTmp);
      T HT = splat<T>(0.0);
      if (ThickOn)
         HT -= HDivTmp;
      if (!EPI || E.StoreTend)
         stnt<T>(HTend, OffS, HT);
      if (EPI)
         stageUpdate<T, true>(E, OffS, HT, Hs);
      if (DoDel2Tr) {
         const size_t CStride = (size_t)M.NCellsSize * K;
#pragma nounroll
         for (int Lt = 0; Lt < NT; ++Lt) {
            loopFence();
            const Real *TrL = uniformPtr(Tr + Lt * CStride);
            T Tn[TME];
#pragma unroll
            for (int J = 0; J < TME; ++J)
               Tn[J] = ldo<T>(TrL, OffN[J]);
            const T Ts = ldo<T>(TrL, OffS);
            T Tmp      = splat<T>(0.0);
#pragma unroll
            for (int J = 0; J < TME; ++J) {
               // Fast: the staged coefficient carries the orientation, (T1-T0) == +-(Tn-Ts) exactly
               const T Grad = Fast ? T(Tn[J] - Ts) : T(pick(IsC0[J], Tn[J], Ts) - pick(IsC0[J], Ts, Tn[J]));
               Tmp -= L.D2T[Le * TME + J] * HMeanJ[J] * Grad;
            }
            stnt<T>(uniformPtr(Del2Tr + Lt * CStride), OffS, Tmp * InvA);
         }
      }
   }
};

// ---------------------------------------------------------------------------------------
// L1 cell pass with the vertex pass and the side-0 PV sums folded in (HorzMesh.h: CellL1OK).
//
// The thread of (cell, levels) already holds h at the cell and its neighbours and u on its edges.  With u on the
// TME "spoke" edges (the edges between consecutive neighbours) it evaluates VorticityAuxVars::computeVarsOnVertex
// (VorticityAuxVars.h:24-59) at ALL its ring vertices -- the vertex's own coefficients, added in the vertex's own
// slot order, so each value has the bits the vertex kernel produces, whichever of the three cells around the
// vertex computes it -- stores the vertices it owns (RelVort, 1/LayerThickVertex), and goes straight on to the
// side-0 half of PotentialVortHAdvOnEdge (CellPVBody<.., 0>) with the normalised vorticities still in registers.
// Against VortVertexBody + FusedCell1Body + CellPVBody<side 0> this reads h and u once instead of three times and
// never re-reads the two vertex arrays: 96 B per cell-level less HBM traffic and two launches less.
/// INLO: cells with NR - 1 edges (the pentagons of a hexagon mesh) do their side-0 sums here as well, with the ring code
/// instantiated a second time, instead of through a list launch of CellPVBody (12 pentagons on a QU240-sized sphere:
/// that launch was 9 % of the RHS).  Only instantiated for meshes that have such cells.
/// FL (compile-time list / width flags of the cell bodies): bit 0 = the launch may run over a cell list (`List`), bit 1 =
/// the mesh has cells with more edges than these tables hold (narrow view: such cells are skipped here and served by a
/// list launch on the wide tables).  The full sweeps of a mesh without wider cells are instantiated with FL = 0: no list
/// selects, no width test -- the instruction stream they had before lists and narrow tables existed.
template <int TME, bool Fast, bool EPI = false, int NR = TME, bool INLO = false, int FL = 3> struct FusedCellL1PVBody {
   __device__ __forceinline__ int cellOf(int I) const {
      if constexpr ((FL & 1) != 0)
         return List ? List[I] : I;
      else
         return I;
   }
   static constexpr int MinWaves = OMEGA_CELL_MINW;
   static constexpr int MaxW     = OMEGA_CELL_MAXW;
   /// tile-local index opaque per chunk (KernelCommon.h: tileKernel): 240 -> 174 VGPRs at TME = 6, this launch - 2 %.
   /// (Three waves per SIMD are then within reach -- 166 VGPRs with MinWaves = 3 -- and measured slower: + 4 % planar,
   /// + 21 % with the inlined pentagon code spilling; the level-3 bodies lose 4 % with the opaque index.)
   static constexpr bool OpaqueLe = true;
   static constexpr int TM1       = TME - 1;
   MeshView M;
   int K, NT;
   TendParams P;
   int DoDel2Tr;
   const Real *H, *U, *Tr;
   Real *KE, *Div, *HTend, *Del2Tr, *RelVortV, *InvThickV, *Partial;
   StageEpi E{}; // thickness stage update (EPI)
   const int *List = nullptr; // optional cell list (the wide cells of a mesh with narrow tables: launchFusedT)
   int SkipBad     = 0;       // the mesh has cells outside the ring tables (MeshView::BadCells): skipped here
   struct Lds {
      Real *KEC, *DivC, *DvS, *D2T, *InvA, *Wt, *FV, *KC, *VC;
      int *Edge, *NbrF, *Spoke, *Sel, *Ring, *Role, *N;
   };
   size_t ldsBytes(int Tile) const {
      return ldsRound8(sizeof(Real) * Tile * TME) * 5 + ldsRound8(sizeof(Real) * Tile) +
             ldsRound8(sizeof(Real) * Tile * TME * TM1) + ldsRound8(sizeof(Real) * Tile * TME * 3) * 2 +
             ldsRound8(sizeof(int) * Tile * TME) * 6 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.KEC   = C.take<Real>(Tile * TME);
      L.DivC  = C.take<Real>(Tile * TME);
      L.DvS   = C.take<Real>(Tile * TME);
      L.D2T   = C.take<Real>(Tile * TME);
      L.FV    = C.take<Real>(Tile * TME);
      L.InvA  = C.take<Real>(Tile);
      L.Wt    = C.take<Real>(Tile * TME * TM1);
      L.KC    = C.take<Real>(Tile * TME * 3);
      L.VC    = C.take<Real>(Tile * TME * 3);
      L.Edge  = C.take<int>(Tile * TME);
      L.NbrF  = C.take<int>(Tile * TME);
      L.Spoke = C.take<int>(Tile * TME);
      L.Sel   = C.take<int>(Tile * TME);
      L.Ring  = C.take<int>(Tile * TME);
      L.Role  = C.take<int>(Tile * TME);
      L.N     = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt * TME; I += NThr) {
         const int Cl = I / TME, Jl = I - Cl * TME;
         const int C  = cellOf(First + Cl);
         const size_t G = (size_t)C * TME + Jl;
         L.KEC[I]       = M.KECoefOnCell[G];
         L.DivC[I]      = M.DivCoefOnCell[G];
         L.DvS[I]       = M.DvSignOnCell[G];
         L.D2T[I]       = Fast ? M.Del2TrCoefSOnCell[G] : M.Del2TrCoefOnCell[G];
         // slot N of a cell with N < TME edges repeats slot 0 (its coefficients are zero, so it adds exact zeros to the
         // cell sums) -- the ring code below then finds "the slot after N-1" without a wrap-around select
         const size_t G0 = (Jl == M.NEdgesOnCell[C]) ? (size_t)C * TME : G;
         L.Edge[I]      = M.EdgesOnCell[G0];
         L.NbrF[I]      = M.NbrFlagOnCell[G0];
         L.Spoke[I]     = M.SpokeOnCell[G];
         L.Sel[I]       = M.VortSelOnCell[G];
         L.Ring[I]      = M.VertRingOnCell[G];
         L.FV[I]        = M.FVertex[M.VertRingOnCell[G]];
         L.Role[I]      = M.PVRoleOnCell[G];
      }
      if (!(FL & 1) || !List) { // a sweep: the tile's rows are one contiguous piece of every table
         for (int I = Tid; I < Cnt * TME * TM1; I += NThr)
            L.Wt[I] = M.PVWeightOnCell[(size_t)First * TME * TM1 + I];
         for (int I = Tid; I < Cnt * TME * 3; I += NThr) {
            L.KC[I] = M.KiteCoefOnCell[(size_t)First * TME * 3 + I];
            L.VC[I] = M.VortCoefOnCell[(size_t)First * TME * 3 + I];
         }
      } else {
         for (int I = Tid; I < Cnt * TME * TM1; I += NThr) {
            const int Cl = I / (TME * TM1);
            L.Wt[I]      = M.PVWeightOnCell[(size_t)List[First + Cl] * TME * TM1 + (I - Cl * TME * TM1)];
         }
         for (int I = Tid; I < Cnt * TME * 3; I += NThr) {
            const int Cl   = I / (TME * 3);
            const size_t G = (size_t)List[First + Cl] * TME * 3 + (I - Cl * TME * 3);
            L.KC[I]        = M.KiteCoefOnCell[G];
            L.VC[I]        = M.VortCoefOnCell[G];
         }
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         const int C = cellOf(First + I);
         L.InvA[I]   = M.InvAreaCell[C];
         L.N[I]      = M.NEdgesOnCellRing[C]; // (99 for a cell outside the ring tables)
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      const int N = L.N[Le];
      if (((FL & 2) != 0 || SkipBad) && N > TME)
         return; // a cell wider than these tables, or outside the ring tables: it has its own (list) launch
      const int ICell       = cellOf(IElem);
      const bool FluxUpwind = Fast ? false : (P.FluxThicknessUpwind != 0);
      const bool ThickOn    = Fast ? true : (P.ThicknessFluxTendencyEnable != 0);
      const Real InvA       = L.InvA[Le];
      const unsigned OffS   = rowOff<T>(ICell, K, Kv);
      unsigned OffN[TME], OffE[TME];
      bool IsC0[TME];
      T Ue[TME], Hn[TME], Usp[TME];
#pragma unroll
      for (int J = 0; J < TME; ++J) {
         const int F = L.NbrF[Le * TME + J];
         OffN[J]     = rowOff<T>(F & 0x3fffffff, K, Kv);
         IsC0[J]     = (F >> 30) != 0;
         OffE[J]     = rowOff<T>(L.Edge[Le * TME + J], K, Kv);
         Ue[J]       = ldo<T>(U, OffE[J]);
         Hn[J]       = ldo<T>(H, OffN[J]);
         Usp[J]      = ldo<T>(U, rowOff<T>(L.Spoke[Le * TME + J], K, Kv));
      }
      const T Hs = ldo<T>(H, OffS);
      StagePre<T> PreH{};
      if (EPI)
         PreH = stagePre<T, true>(E, OffS);
      // ---- KineticAuxVars / ThicknessFluxDivOnCell: exactly FusedCell1Body ----
      T HMeanJ[TME], Flux[TME];
      {
         T KETmp = splat<T>(0.0), DivTmp = splat<T>(0.0), HDivTmp = splat<T>(0.0);
#pragma unroll
         for (int J = 0; J < TME; ++J) {
            const T Mean = 0.5 * (Hs + Hn[J]); // 0.5*(h(c0)+h(c1)): a+b == b+a
            HMeanJ[J]    = Mean;
            Flux[J]      = Mean;
            if (FluxUpwind)
               Flux[J] = upwind(Ue[J], pick(IsC0[J], Hs, Hn[J]), pick(IsC0[J], Hn[J], Hs));
            KETmp += L.KEC[Le * TME + J] * Ue[J] * Ue[J];
            DivTmp -= L.DivC[Le * TME + J] * Ue[J];
            HDivTmp -= L.DvS[Le * TME + J] * Flux[J] * Ue[J] * InvA;
         }
         stnt<T>(KE, OffS, KETmp);
         stnt<T>(Div, OffS, DivTmp);
         T HT = splat<T>(0.0);
         if (ThickOn)
            HT -= HDivTmp;
         if (!EPI || E.StoreTend)
            stnt<T>(HTend, OffS, HT);
         if (EPI)
            stageApply<T, true>(E, OffS, HT, Hs, PreH);
      }
      // ---- VorticityAuxVars::computeVarsOnVertex at every ring vertex (VorticityAuxVars.h:24-59) ----
      // ((0 + t0) + t1) + t2 in the vertex's slot order: t0 + t1 commutes, so with the coefficients staged per
      // role only the role of the last slot has to be selected.  Slot N of a cell with N < TME edges repeats slot 0
      // (stage()), so "slot R+1" needs no wrap-around select.
      T QR[TME], QF[TME];
      {
         const T Zero = splat<T>(0.0);
#pragma unroll
         for (int R = 0; R < TME; ++R) {
            const int R1  = (R + 1) % TME;
            const int Sel = L.Sel[Le * TME + R];
            const int Lc = Sel & 3, Lu = (Sel >> 2) & 3;
            const T PA = L.KC[(Le * TME + R) * 3 + 0] * Hs, PB = L.KC[(Le * TME + R) * 3 + 1] * Hn[R],
                    PC = L.KC[(Le * TME + R) * 3 + 2] * Hn[R1];
            const T X  = pick(Lc == 0, PB, PA), Y = pick(Lc == 2, PB, PC), Z = pick(Lc == 0, PA, pick(Lc == 1, PB, PC));
            const T LayerThickVertex = ((Zero + X) + Y) + Z;
            const T UA = L.VC[(Le * TME + R) * 3 + 0] * Ue[R], UB = L.VC[(Le * TME + R) * 3 + 1] * Ue[R1],
                    UC = L.VC[(Le * TME + R) * 3 + 2] * Usp[R];
            const T Xu = pick(Lu == 0, UB, UA), Yu = pick(Lu == 2, UB, UC), Zu = pick(Lu == 0, UA, pick(Lu == 1, UB, UC));
            const T RelVortTmp = ((Zero + Xu) + Yu) + Zu;
            const T Inv        = 1. / LayerThickVertex;
            { // the cell that stores the vertex (bit 4) writes; the other lanes' stores are switched off, not branched around
               const bool Own      = (Sel >> 4) & 1;
               const unsigned OffV = rowOff<T>(L.Ring[Le * TME + R], K, Kv);
               stoIf<T>(Own, RelVortV, OffV, RelVortTmp);
               stoIf<T>(Own, InvThickV, OffV, Inv);
            }
            QR[R] = RelVortTmp * Inv;         // NormRelVortVertex   (:50-51)
            QF[R] = L.FV[Le * TME + R] * Inv; // NormPlanetVortVertex (:52-53)
         }
      }
      // ---- side-0 half of PotentialVortHAdvOnEdge: exactly CellPVBody<TME, Fast, 0, NR> ----
      // (NR = the valence of most cells: MaxEdges, or MaxEdges-1 on a mesh of hexagons with a few heptagons; the other
      // valences go through the list launches of CellPVBody)
      auto Side0 = [&](auto RingSize) {
         constexpr int NRr = decltype(RingSize)::value; // this cell's valence; table strides stay TME
         bool Any          = false;
#pragma unroll
         for (int J = 0; J < NRr; ++J)
            Any |= L.Role[Le * TME + J] == 1;
         if (Any) {
            T QRe[TME], QFe[TME];
#pragma unroll
            for (int J = 0; J < NRr; ++J) {
               const int Jm = (J + NRr - 1) % NRr;
               QRe[J]       = 0.5 * (QR[Jm] + QR[J]);
               QFe[J]       = 0.5 * (QF[Jm] + QF[J]);
            }
#pragma unroll
            for (int I = 0; I < NRr; ++I) {
               if (L.Role[Le * TME + I] != 1)
                  continue;
               T Acc = splat<T>(0.0);
#pragma unroll
               for (int J = 1; J < NRr; ++J) {
                  const int Kk     = (I + J) % NRr;
                  const T NormVort = (QRe[I] + QFe[I] + QRe[Kk] + QFe[Kk]) * 0.5;
                  Acc += L.Wt[(Le * TME + I) * TM1 + J - 1] * Flux[Kk] * Ue[Kk] * NormVort;
               }
               sto<T>(Partial, OffE[I], Acc);
            }
         }
      };
      if (N == NR) {
         Side0(std::integral_constant<int, NR>{});
      } else if constexpr (INLO && NR >= 5) {
         if (N == NR - 1)
            Side0(std::integral_constant<int, NR - 1>{});
      }
      // ---- TracerAuxVars::computeVarsOnCells: exactly FusedCell1Body ----
      if (DoDel2Tr) {
         const size_t CStride = (size_t)M.NCellsSize * K;
         // TU tracers per trip: their gathers are asked for together (one memory round trip per trip); a tracer past
         // the last one has its accesses switched off.  Three per trip where that divides the tracer count and the
         // registers are there (6-slot tables: 174 VGPRs either way; level 1 -0.6 ... -2 %), two otherwise (also for the
         // 7-wide tables, where it costs 32 B of scratch: -6 %)
         auto TrLoop = [&](auto Unroll) {
            constexpr int TU = decltype(Unroll)::value;
#pragma nounroll
            for (int Lt = 0; Lt < NT; Lt += TU) {
               loopFence();
               T Tn[TU][TME], Ts[TU];
#pragma unroll
               for (int Q = 0; Q < TU; ++Q) {
                  const bool Valid = TU == 1 || Lt + Q < NT;
                  const Real *TrL  = uniformPtr(Tr + (Valid ? Lt + Q : Lt) * CStride);
#pragma unroll
                  for (int J = 0; J < TME; ++J)
                     Tn[Q][J] = ldoIf<T>(Valid, TrL, OffN[J]);
                  Ts[Q] = ldoIf<T>(Valid, TrL, OffS);
               }
#pragma unroll
               for (int Q = 0; Q < TU; ++Q) {
                  const bool Valid = TU == 1 || Lt + Q < NT;
                  T Tmp            = splat<T>(0.0);
#pragma unroll
                  for (int J = 0; J < TME; ++J) {
                     const T Grad =
                         Fast ? T(Tn[Q][J] - Ts[Q]) : T(pick(IsC0[J], Tn[Q][J], Ts[Q]) - pick(IsC0[J], Ts[Q], Tn[Q][J]));
                     Tmp -= L.D2T[Le * TME + J] * HMeanJ[J] * Grad;
                  }
                  stntIf<T>(Valid, uniformPtr(Del2Tr + (Valid ? Lt + Q : Lt) * CStride), OffS, Tmp * InvA);
               }
            }
         };
         if constexpr (TME <= 6 && Fast) {
            if (NT % 3 == 0) {
               TrLoop(std::integral_constant<int, 3>{});
               return;
            }
         }
         TrLoop(std::integral_constant<int, 2>{});
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
   const int *List = nullptr; // optional cell list (the cells outside the ring tables: MeshView::BadCells)
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
         size_t G = (size_t)First * ME + I;
         if (List) {
            const int Cl = I / ME;
            G            = (size_t)List[First + Cl] * ME + (I - Cl * ME);
         }
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
         L.N[I] = M.NEdgesOnCell[List ? List[First + I] : First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      const int ME    = M.MaxEdges;
      const int ICell = List ? List[IElem] : IElem;
      const int N     = L.N[Le];
      T Tmp           = splat<T>(0.0);
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

// L2 cell pass, ring form (HorzMesh::buildDel2Tables): same arithmetic as FusedDel2CellBody with
// every row gathered once -- Div at the cell and its TME neighbours, RelVort on its TME ring vertices.
/// (FL only names the instantiation here: this body keeps its run-time list / width tests -- with them compiled out the
/// three-sweep launch of a mesh with narrow tables came out 23 % slower, 224 -> 276 us on the Fibonacci sphere)
template <int TME, int FL = 3> struct Del2CellRingBody {
   static constexpr bool HoistTables = true; // (KernelCommon.h: measured 0.588 against 0.600 ms for the pair)
   MeshView M;
   int K;
   const Real *Div, *RelVort;
   Real *Del2Div;
   const int *List = nullptr; // optional cell list (the wide cells of a mesh with narrow tables: launchFusedT)
   struct Lds {
      Real *DivC, *InvDc, *GradS, *CurlC;
      int *Nbr, *Ring, *N;
   };
   size_t ldsBytes(int Tile) const {
      return ldsRound8(sizeof(Real) * Tile * TME) * 4 + ldsRound8(sizeof(int) * Tile * TME) * 2 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.DivC  = C.take<Real>(Tile * TME);
      L.InvDc = C.take<Real>(Tile * TME);
      L.GradS = C.take<Real>(Tile * TME);
      L.CurlC = C.take<Real>(Tile * TME);
      L.Nbr   = C.take<int>(Tile * TME);
      L.Ring  = C.take<int>(Tile * TME);
      L.N     = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt * TME; I += NThr) {
         size_t G = (size_t)First * TME + I;
         if (List) {
            const int Cl = I / TME;
            G            = (size_t)List[First + Cl] * TME + (I - Cl * TME);
         }
         L.DivC[I]      = M.DivCoefOnCell[G];
         L.InvDc[I]     = M.InvDcOnCell[G];
         L.GradS[I]     = M.Del2GradMaskSOnCell[G];
         L.CurlC[I]     = M.Del2CurlCoefOnCell[G];
         L.Nbr[I]       = M.NbrFlagOnCell[G] & 0x3fffffff;
         L.Ring[I]      = M.VertRingOnCell[G];
      }
      for (int I = Tid; I < Cnt; I += NThr)
         L.N[I] = M.NEdgesOnCellRing[List ? List[First + I] : First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      if (L.N[Le] > TME)
         return; // a cell wider than these tables, or outside the ring tables (99): it has its own (list) launch
      const int ICell = List ? List[IElem] : IElem;
      T Dn[TME], Rv[TME];
#pragma unroll
      for (int J = 0; J < TME; ++J) {
         Dn[J] = ldo<T>(Div, rowOff<T>(L.Nbr[Le * TME + J], K, Kv));
         Rv[J] = ldo<T>(RelVort, rowOff<T>(L.Ring[Le * TME + J], K, Kv));
      }
      const unsigned OffS = rowOff<T>(ICell, K, Kv);
      const T Ds          = ldo<T>(Div, OffS);
      T Tmp               = splat<T>(0.0);
#pragma unroll
      for (int J = 0; J < TME; ++J) {
         const int I      = Le * TME + J;
         const int Jm     = (J + TME - 1) % TME;
         const T GradDiv  = (Dn[J] - Ds) * L.InvDc[I];     // x orientation, folded into GradS
         const T CurlVort = (Rv[J] - Rv[Jm]) * L.CurlC[I]; // -(RelVort(v1) - RelVort(v0)) * InvDvEdgeDel2
         const T Del2E    = L.GradS[I] * GradDiv + CurlVort;
         Tmp -= L.DivC[I] * Del2E;
      }
      stnt<T>(Del2Div, OffS, Tmp);
   }
};

// L2 vertex pass for VertexDegree 3, each row gathered once (7 instead of 12).
struct Del2VertexSelBody {
   static constexpr bool HoistTables = true;
   MeshView M;
   int K;
   const Real *Div, *RelVort;
   Real *Del2RelVort;
   struct Lds {
      Real *VortC, *InvDc, *Mask, *CurlC;
      int *Cell, *NbrV, *Sel;
   };
   size_t ldsBytes(int Tile) const { return ldsRound8(sizeof(Real) * Tile * 3) * 4 + ldsRound8(sizeof(int) * Tile * 3) * 3; }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.VortC = C.take<Real>(Tile * 3);
      L.InvDc = C.take<Real>(Tile * 3);
      L.Mask  = C.take<Real>(Tile * 3);
      L.CurlC = C.take<Real>(Tile * 3);
      L.Cell  = C.take<int>(Tile * 3);
      L.NbrV  = C.take<int>(Tile * 3);
      L.Sel   = C.take<int>(Tile * 3);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt * 3; I += NThr) {
         const size_t G = (size_t)First * 3 + I;
         L.VortC[I]     = M.VortCoefOnVertex[G];
         L.InvDc[I]     = M.InvDcOnVertex[G];
         L.Mask[I]      = M.Del2MaskOnVertex[G];
         L.CurlC[I]     = M.Del2CurlCoefOnVertex[G];
         L.Cell[I]      = M.CellsOnVertex[G];
         L.NbrV[I]      = M.NbrVertOnVertex[G];
         L.Sel[I]       = M.Del2SelOnVertex[G];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IVertex, int Kv) const {
      T D[3], Rn[3];
#pragma unroll
      for (int J = 0; J < 3; ++J) {
         D[J]  = ldo<T>(Div, rowOff<T>(L.Cell[Le * 3 + J], K, Kv));
         Rn[J] = ldo<T>(RelVort, rowOff<T>(L.NbrV[Le * 3 + J], K, Kv));
      }
      const unsigned OffS = rowOff<T>(IVertex, K, Kv);
      const T Rs          = ldo<T>(RelVort, OffS);
      T Tmp               = splat<T>(0.0);
#pragma unroll
      for (int J = 0; J < 3; ++J) {
         const int I  = Le * 3 + J;
         const int S  = L.Sel[I];
         const int S0 = S & 3, S1 = S >> 2;
         const T D0   = pick(S0 == 0, D[0], pick(S0 == 1, D[1], D[2]));
         const T D1   = pick(S1 == 0, D[0], pick(S1 == 1, D[1], D[2]));
         const T GradDiv  = (D1 - D0) * L.InvDc[I];
         const T CurlVort = (Rn[J] - Rs) * L.CurlC[I];
         const T Del2E    = L.Mask[I] * GradDiv + CurlVort;
         Tmp += L.VortC[I] * Del2E;
      }
      stnt<T>(Del2RelVort, OffS, Tmp);
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
   int KLog = 0; ///< number of levels (K is the row pitch): set by launchTile
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
      if (P.BottomDragTendencyEnable && (Kv + 1) * W >= KLog) {
         const int KBot          = KLog - 1;
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
// L3 edge pass: every velocity term (TendencyTerms.h:81-334) in registers.  The edge-located
// inputs of PotentialVortHAdvOnEdge at each EdgesOnEdge neighbour (FluxLayerThickEdge,
// NormRelVortEdge, NormPlanetVortEdge) are rebuilt from h at its two cells and the normalised
// vorticities at its two vertices (LayerThicknessAuxVars.h:25-61, VorticityAuxVars.h:61-76);
// SshCell from h - BottomDepth (LayerThicknessAuxVars.h:63-82).
//
// The stencil is walked in chain form (HorzMesh.h PVChain*): the other edges of each of the two
// cells of the edge, in EdgesOnEdge order, share end vertices with their successors and all have
// that cell as one of their two cells, so per side only the chain's vertices and the far cells
// are gathered (a+b == b+a exactly, so which end vertex / cell comes first does not matter for
// the means; the upwind choice keeps its flag).
template <int TME, bool Fast, bool EPI = false, bool INV = false> struct FusedEdgeChainBody {
   static constexpr int MinWaves = OMEGA_EDGE_MINW;
   static constexpr int MaxW     = OMEGA_EDGE_MAXW;
   static constexpr int TM1      = TME - 1;
   MeshView M;
   int K;
   TendParams P;
   const Real *H, *U;
   /// INV (the irregular-edge list launch next to the cell-centric kernels): VortA = InvThickVertex and the
   /// normalised vorticities are rebuilt as the vertex kernel computes them (VorticityAuxVars.h:50-53);
   /// otherwise (this kernel as the whole edge pass) VortA / VortB = NormRelVortVertex / NormPlanetVortVertex
   const Real *RelVort, *VortA, *VortB, *KE, *Div, *Del2Div, *Del2RelVort, *NormalStress;
   template <class T> __device__ __forceinline__ void normVort(Real FV, unsigned Off, T &QR, T &QF) const {
      if (INV) {
         const T Iv = ldo<T>(VortA, Off);
         QR         = ldo<T>(RelVort, Off) * Iv;
         QF         = FV * Iv;
      } else {
         QR = ldo<T>(VortA, Off);
         QF = ldo<T>(VortB, Off);
      }
   }
   Real *Tend;
   const I4 *EdgeList = nullptr; ///< if set, element i of the sweep is edge EdgeList[i]
   StageEpi E{};                 ///< velocity stage update (EPI)
   int KLog = 0;                 ///< number of levels (K is the row pitch): set by launchTile
   __device__ int edgeOf(int I) const { return EdgeList ? EdgeList[I] : I; }
   struct Lds {
      Real *W, *InvDc, *InvDv, *Mask, *MaskGrav, *C2, *C4, *BD0, *BD1, *FCh, *F0, *F1; // F*: FVertex of ChV / V0 / V1
      int *ChV, *ChF, *ChE, *C0, *C1, *V0, *V1;
   };
   size_t ldsBytes(int Tile) const {
      return ldsRound8(sizeof(Real) * Tile * 2 * TM1) + ldsRound8(sizeof(Real) * Tile) * 10 +
             ldsRound8(sizeof(Real) * Tile * 2 * TME) +
             ldsRound8(sizeof(int) * Tile * 2 * TME) + ldsRound8(sizeof(int) * Tile * 2 * TM1) * 2 +
             ldsRound8(sizeof(int) * Tile) * 4;
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.W        = C.take<Real>(Tile * 2 * TM1);
      L.InvDc    = C.take<Real>(Tile);
      L.InvDv    = C.take<Real>(Tile);
      L.Mask     = C.take<Real>(Tile);
      L.MaskGrav = C.take<Real>(Tile);
      L.C2       = C.take<Real>(Tile);
      L.C4       = C.take<Real>(Tile);
      L.BD0      = C.take<Real>(Tile);
      L.BD1      = C.take<Real>(Tile);
      L.F0       = C.take<Real>(Tile);
      L.F1       = C.take<Real>(Tile);
      L.FCh      = C.take<Real>(Tile * 2 * TME);
      L.ChV      = C.take<int>(Tile * 2 * TME);
      L.ChF      = C.take<int>(Tile * 2 * TM1);
      L.ChE      = C.take<int>(Tile * 2 * TM1);
      L.C0       = C.take<int>(Tile);
      L.C1       = C.take<int>(Tile);
      L.V0       = C.take<int>(Tile);
      L.V1       = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const Real Grav = 9.80665; // TendencyTerms.h:176
      for (int I = Tid; I < Cnt * 2 * TM1; I += NThr) {
         const int Ei   = I / (2 * TM1);
         const size_t G = (size_t)edgeOf(First + Ei) * 2 * TM1 + (I - Ei * 2 * TM1);
         L.W[I]         = M.PVChainWeight[G];
         L.ChF[I]       = M.PVChainFar[G];
         L.ChE[I]       = M.PVChainEdge[G];
      }
      for (int I = Tid; I < Cnt * 2 * TME; I += NThr) {
         const int Ei = I / (2 * TME);
         L.ChV[I]     = M.PVChainVert[(size_t)edgeOf(First + Ei) * 2 * TME + (I - Ei * 2 * TME)];
         L.FCh[I]     = M.FVertex[L.ChV[I]];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         const int E     = edgeOf(First + I);
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
         L.F0[I]       = M.FVertex[M.VerticesOnEdge[2 * E]];
         L.F1[I]       = M.FVertex[M.VerticesOnEdge[2 * E + 1]];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      const int IEdge       = edgeOf(IElem);
      const bool FluxUpwind = Fast ? false : (P.FluxThicknessUpwind != 0);
      const bool PVOn = Fast ? true : (P.PVTendencyEnable != 0), KEOn = Fast ? true : (P.KETendencyEnable != 0);
      const bool SSHOn = Fast ? true : (P.SSHTendencyEnable != 0), D2On = Fast ? true : (P.VelDiffTendencyEnable != 0);
      const bool D4On   = Fast ? true : (P.VelHyperDiffTendencyEnable != 0);
      const bool WindOn = Fast ? false : (P.WindForcingTendencyEnable != 0);
      const bool DragOn = Fast ? false : (P.BottomDragTendencyEnable != 0);
      const unsigned OffC0 = rowOff<T>(L.C0[Le], K, Kv), OffC1 = rowOff<T>(L.C1[Le], K, Kv);
      const unsigned OffV0 = rowOff<T>(L.V0[Le], K, Kv), OffV1 = rowOff<T>(L.V1[Le], K, Kv);
      const Real InvDc = L.InvDc[Le], InvDv = L.InvDv[Le];
      const T H0 = ldo<T>(H, OffC0), H1 = ldo<T>(H, OffC1);
      T TendV = splat<T>(0.0);
      if (PVOn) {
         // NormRelVortEdge / NormPlanetVortEdge of this edge (VorticityAuxVars.h:68-74)
         T QR0, QF0, QR1, QF1;
         normVort<T>(L.F0[Le], OffV0, QR0, QF0);
         normVort<T>(L.F1[Le], OffV1, QR1, QF1);
         const T QRe = 0.5 * (QR0 + QR1);
         const T QFe = 0.5 * (QF0 + QF1);
         T VortTmp   = splat<T>(0.0);
#pragma unroll
         for (int Sd = 0; Sd < 2; ++Sd) {
            const T Hs    = Sd == 0 ? H0 : H1;
            const int BV  = (Le * 2 + Sd) * TME, BM = (Le * 2 + Sd) * TM1;
            T QR[TME], QF[TME], Uj[TM1], Hf[TM1];
            bool First[TM1];
#pragma unroll
            for (int J = 0; J < TME; ++J) {
               const unsigned Off = rowOff<T>(L.ChV[BV + J], K, Kv);
               normVort<T>(L.FCh[BV + J], Off, QR[J], QF[J]);
            }
#pragma unroll
            for (int J = 0; J < TM1; ++J) {
               const int F = L.ChF[BM + J];
               First[J]    = (F >> 30) != 0;
               Uj[J]       = ldo<T>(U, rowOff<T>(L.ChE[BM + J], K, Kv));
               Hf[J]       = ldo<T>(H, rowOff<T>(F & 0x3fffffff, K, Kv));
            }
#pragma unroll
            for (int J = 0; J < TM1; ++J) {
               T Flux = 0.5 * (Hs + Hf[J]);
               if (FluxUpwind)
                  Flux = upwind(Uj[J], pick(First[J], Hs, Hf[J]), pick(First[J], Hf[J], Hs));
               const T QRj  = 0.5 * (QR[J] + QR[J + 1]);
               const T QFj  = 0.5 * (QF[J] + QF[J + 1]);
               const T NormVort = (QRe + QFe + QRj + QFj) * 0.5;
               VortTmp += L.W[BM + J] * Flux * Uj[J] * NormVort;
            }
         }
         TendV += L.Mask[Le] * VortTmp;
      }
      if (KEOn)
         TendV -= L.Mask[Le] * (ldo<T>(KE, OffC1) - ldo<T>(KE, OffC0)) * InvDc;
      if (SSHOn) {
         const T Ssh0 = H0 - L.BD0[Le], Ssh1 = H1 - L.BD1[Le];
         TendV -= L.MaskGrav[Le] * (Ssh1 - Ssh0) * InvDc;
      }
      if (D2On) {
         const T Del2U = ((ldo<T>(Div, OffC1) - ldo<T>(Div, OffC0)) * InvDc -
                          (ldo<T>(RelVort, OffV1) - ldo<T>(RelVort, OffV0)) * InvDv);
         TendV += L.C2[Le] * Del2U;
      }
      if (D4On) {
         const T Del2U = (P.DivFactor * (ldo<T>(Del2Div, OffC1) - ldo<T>(Del2Div, OffC0)) * InvDc -
                          (ldo<T>(Del2RelVort, OffV1) - ldo<T>(Del2RelVort, OffV0)) * InvDv);
         TendV -= L.C4[Le] * Del2U;
      }
      constexpr int W = VecW<T>::W;
      if (WindOn && Kv == 0) {
         const Real HMean0       = 0.5 * (getc(H0, 0) + getc(H1, 0));
         const Real InvThickEdge = 1. / HMean0;
         setc(TendV, 0, getc(TendV, 0) + L.Mask[Le] * InvThickEdge * NormalStress[IEdge] / P.Density0);
      }
      if (DragOn && (Kv + 1) * W >= KLog) {
         const int KBot          = KLog - 1;
         const int Comp          = KBot - Kv * W;
         const Real VelNormEdge  = sqrt(KE[(size_t)L.C0[Le] * K + KBot] + KE[(size_t)L.C1[Le] * K + KBot]);
         const Real HMeanB       = 0.5 * (getc(H0, Comp) + getc(H1, Comp));
         const Real InvThickEdge = 1. / HMeanB;
         setc(TendV, Comp,
              getc(TendV, Comp) - L.Mask[Le] * P.BottomDragCoeff * VelNormEdge * InvThickEdge * U[(size_t)IEdge * K + KBot]);
      }
      const unsigned OffT = rowOff<T>(IEdge, K, Kv);
      if (!EPI || E.StoreTend)
         sto<T>(Tend, OffT, TendV);
      if (EPI)
         stageUpdate<T, false>(E, OffT, TendV, TendV);
   }
};

// ---------------------------------------------------------------------------------------
// PotentialVortHAdvOnEdge (TendencyTerms.h:81-108), cell-centric.  For a regular edge (HorzMesh.h
// CellPV) the reference's sum runs first over the other edges of cell 0, then over the other edges
// of cell 1.  Everything the cell-s part needs lives on the ring of cell s, so a thread owning
// (cell, levels) gathers the ring once -- u on its ME edges, h on its ME neighbours and itself,
// NormRelVort / NormPlanetVort on its ME vertices: 4*ME+1 gathers -- and produces the partial sums
// of ALL its edges (ME*(ME-1) terms), instead of 7..11 gathers per single term in the edge-centric
// form.  Side = 0 launches first and stores the running sums; Side = 1 continues each sum from the
// stored value, so the additions happen in exactly the reference's order.
template <int TME, bool Fast, int Side, int NR = TME> struct CellPVBody {
   static constexpr int MinWaves = OMEGA_CELL_MINW;
   static constexpr int TM1      = TME - 1;
   MeshView M;
   int K;
   TendParams P;
   const Real *H, *U, *RelVortV, *InvThickV; // NormRelVort = RelVort*InvThick, NormPlanetVort = FVertex*InvThick
   Real *Partial; // [NEdgesSize][K] running PV sums
   const int *List = nullptr; // optional cell list (the launches for the rarer valences)
   struct Lds {
      Real *Wt, *FV;
      int *Edge, *NbrF, *Ring, *Role, *N;
   };
   __host__ __device__ size_t ldsBytes(int Tile) const {
      return ldsRound8(sizeof(Real) * Tile * TME * TM1) + ldsRound8(sizeof(Real) * Tile * TME) +
             ldsRound8(sizeof(int) * Tile * TME) * 4 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.Wt   = C.take<Real>(Tile * TME * TM1);
      L.FV   = C.take<Real>(Tile * TME);
      L.Edge = C.take<int>(Tile * TME);
      L.NbrF = C.take<int>(Tile * TME);
      L.Ring = C.take<int>(Tile * TME);
      L.Role = C.take<int>(Tile * TME);
      L.N    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt * TME * TM1; I += NThr) {
         const int Le = I / (TME * TM1);
         const int C  = List ? List[First + Le] : First + Le;
         L.Wt[I]      = M.PVWeightOnCell[(size_t)C * TME * TM1 + (I - Le * TME * TM1)];
      }
      for (int I = Tid; I < Cnt * TME; I += NThr) {
         const int Le   = I / TME;
         const int C    = List ? List[First + Le] : First + Le;
         const size_t G = (size_t)C * TME + (I - Le * TME);
         L.Edge[I]      = M.EdgesOnCell[G];
         L.NbrF[I]      = M.NbrFlagOnCell[G];
         L.Ring[I]      = M.RingVertOnCell[G];
         L.FV[I]        = M.FVertex[M.RingVertOnCell[G]];
         L.Role[I]      = M.PVRoleOnCell[G];
      }
      for (int I = Tid; I < Cnt; I += NThr)
         L.N[I] = M.NEdgesOnCell[List ? List[First + I] : First + I];
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      if (L.N[Le] != NR)
         return; // the other valences have their own (list) launches
      // does this cell own any side-`Side` sum?
      bool Any = false;
#pragma unroll
      for (int J = 0; J < NR; ++J)
         Any |= L.Role[Le * TME + J] == Side + 1;
      if (!Any)
         return;
      constexpr int N = NR; // this launch's valence; table strides stay TME
      const int ICell = List ? List[IElem] : IElem;
      const bool FluxUpwind = Fast ? false : (P.FluxThicknessUpwind != 0);
      unsigned OffE[N];
      T Uj[N], Flux[N], QRe[N], QFe[N];
      {
         T Hn[N], QR[N], QF[N];
         bool IsC0[N];
#pragma unroll
         for (int J = 0; J < N; ++J) {
            const int F = L.NbrF[Le * TME + J];
            IsC0[J]     = (F >> 30) != 0;
            OffE[J]     = rowOff<T>(L.Edge[Le * TME + J], K, Kv);
            Uj[J]       = ldo<T>(U, OffE[J]);
            Hn[J]       = ldo<T>(H, rowOff<T>(F & 0x3fffffff, K, Kv));
            const unsigned OffV = rowOff<T>(L.Ring[Le * TME + J], K, Kv);
            const T Iv          = ldo<T>(InvThickV, OffV);
            QR[J]               = ldo<T>(RelVortV, OffV) * Iv; // VorticityAuxVars.h:50-53
            QF[J]               = L.FV[Le * TME + J] * Iv;
         }
         const T Hs = ldo<T>(H, rowOff<T>(ICell, K, Kv));
#pragma unroll
         for (int J = 0; J < N; ++J) {
            // FluxLayerThickEdge of edge slot J (LayerThicknessAuxVars.h:25-61)
            Flux[J] = 0.5 * (Hs + Hn[J]);
            if (FluxUpwind)
               Flux[J] = upwind(Uj[J], pick(IsC0[J], Hs, Hn[J]), pick(IsC0[J], Hn[J], Hs));
            // NormRelVortEdge / NormPlanetVortEdge of edge slot J: mean over its two end vertices,
            // ring vertices J-1 and J (VorticityAuxVars.h:61-76)
            const int Jm = (J + N - 1) % N;
            QRe[J]       = 0.5 * (QR[Jm] + QR[J]);
            QFe[J]       = 0.5 * (QF[Jm] + QF[J]);
         }
      }
#pragma unroll
      for (int I = 0; I < N; ++I) {
         if (L.Role[Le * TME + I] != Side + 1)
            continue;
         T Acc = splat<T>(0.0);
         if (Side == 1)
            Acc = ldo<T>(Partial, OffE[I]);
#pragma unroll
         for (int J = 1; J < N; ++J) {
            const int Kk     = (I + J) % N;
            const T NormVort = (QRe[I] + QFe[I] + QRe[Kk] + QFe[Kk]) * 0.5;
            Acc += L.Wt[(Le * TME + I) * TM1 + J - 1] * Flux[Kk] * Uj[Kk] * NormVort;
         }
         sto<T>(Partial, OffE[I], Acc);
      }
   }
};

// Side-1 PV pass fused with the remaining velocity terms (default term set).  The cell-1 thread of
// a regular edge finishes the PV sum, so it can go on with KE gradient, SSH gradient, del2 and del4
// (TendencyTerms.h:110-265) in the reference's order and store the finished tendency: the running
// sum is read once and never written back, and the separate edge pass disappears.  Everything the
// extra terms need sits on the same ring: h / KE / Div / Del2Div at this cell and the cell across,
// RelVort / Del2RelVort at ring vertices j-1 and j (orientation folded into InvDvS).
template <int TME, int NR = TME, bool EPI = false> struct CellPVFinalBody {
   static constexpr int MinWaves = OMEGA_PVF_MINW;
   static constexpr int TM1      = TME - 1;
   MeshView M;
   int K;
   TendParams P;
   const Real *H, *U, *RelVortV, *InvThickV, *Partial;
   const Real *RelVort, *KE, *Div, *Del2Div, *Del2RelVort;
   Real *Tend;
   const int *List = nullptr;
   StageEpi E{}; // velocity stage update (EPI)
   struct Lds {
      Real *Wt, *InvDc, *InvDvS, *C2, *C4, *BDn, *BDs, *FV;
      int *Edge, *NbrF, *Ring, *Role, *N;
   };
   size_t ldsBytes(int Tile) const {
      return ldsRound8(sizeof(Real) * Tile * TME * TM1) + ldsRound8(sizeof(Real) * Tile * TME) * 6 +
             ldsRound8(sizeof(Real) * Tile) + ldsRound8(sizeof(int) * Tile * TME) * 4 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.Wt     = C.take<Real>(Tile * TME * TM1);
      L.InvDc  = C.take<Real>(Tile * TME);
      L.InvDvS = C.take<Real>(Tile * TME);
      L.C2     = C.take<Real>(Tile * TME);
      L.C4     = C.take<Real>(Tile * TME);
      L.BDn    = C.take<Real>(Tile * TME);
      L.FV     = C.take<Real>(Tile * TME);
      L.BDs    = C.take<Real>(Tile);
      L.Edge   = C.take<int>(Tile * TME);
      L.NbrF   = C.take<int>(Tile * TME);
      L.Ring   = C.take<int>(Tile * TME);
      L.Role   = C.take<int>(Tile * TME);
      L.N      = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt * TME * TM1; I += NThr) {
         const int Le = I / (TME * TM1);
         const int C  = List ? List[First + Le] : First + Le;
         L.Wt[I]      = M.PVWeightOnCell[(size_t)C * TME * TM1 + (I - Le * TME * TM1)];
      }
      for (int I = Tid; I < Cnt * TME; I += NThr) {
         const int Le    = I / TME;
         const int C     = List ? List[First + Le] : First + Le;
         const size_t G  = (size_t)C * TME + (I - Le * TME);
         const int E     = M.EdgesOnCell[G];
         const int F     = M.NbrFlagOnCell[G];
         const Real Mask = M.EdgeMask1D[E];
         L.Edge[I]       = E;
         L.NbrF[I]       = F;
         L.Ring[I]       = M.RingVertOnCell[G];
         L.FV[I]         = M.FVertex[M.RingVertOnCell[G]];
         L.Role[I]       = M.PVRoleOnCell[G];
         L.InvDc[I]      = M.InvDcEdge[E];
         L.InvDvS[I]     = M.RingSignOnCell[G] * M.InvDvEdge[E];
         L.C2[I]         = Mask * P.ViscDel2 * M.MeshScalingDel2[E];
         L.C4[I]         = Mask * P.ViscDel4 * M.MeshScalingDel4[E];
         L.BDn[I]        = M.BottomDepth[F & 0x3fffffff];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         const int C = List ? List[First + I] : First + I;
         L.BDs[I]    = M.BottomDepth[C];
         L.N[I]      = M.NEdgesOnCell[C];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      if (L.N[Le] != NR)
         return;
      bool Any = false;
#pragma unroll
      for (int J = 0; J < NR; ++J)
         Any |= L.Role[Le * TME + J] == 2;
      if (!Any)
         return;
      constexpr int N = NR; // this launch's valence; table strides stay TME
      const int ICell = List ? List[IElem] : IElem;
      const Real Grav = 9.80665; // TendencyTerms.h:176
      unsigned OffE[N], OffN[N], OffV[N];
      T Uj[N], Flux[N], QRe[N], QFe[N], Hn[N];
      const unsigned OffS = rowOff<T>(ICell, K, Kv);
      const T Hs          = ldo<T>(H, OffS);
      {
         T QR[N], QF[N];
#pragma unroll
         for (int J = 0; J < N; ++J) {
            OffE[J] = rowOff<T>(L.Edge[Le * TME + J], K, Kv);
            OffN[J] = rowOff<T>(L.NbrF[Le * TME + J] & 0x3fffffff, K, Kv);
            OffV[J] = rowOff<T>(L.Ring[Le * TME + J], K, Kv);
            Uj[J]   = ldo<T>(U, OffE[J]);
            Hn[J]   = ldo<T>(H, OffN[J]);
            const T Iv = ldo<T>(InvThickV, OffV[J]);
            QR[J]      = ldo<T>(RelVortV, OffV[J]) * Iv;
            QF[J]      = L.FV[Le * TME + J] * Iv;
         }
#pragma unroll
         for (int J = 0; J < N; ++J) {
            Flux[J]      = 0.5 * (Hs + Hn[J]);
            const int Jm = (J + N - 1) % N;
            QRe[J]       = 0.5 * (QR[Jm] + QR[J]);
            QFe[J]       = 0.5 * (QF[Jm] + QF[J]);
         }
      }
      // (the running sums are asked for together ahead of the per-edge blocks: see CellPVFinalTracerBody)
      T Acc[N];
#pragma unroll
      for (int I = 0; I < N; ++I)
         Acc[I] = ldntIf<T>(L.Role[Le * TME + I] == 2, Partial, OffE[I]);
#pragma unroll
      for (int I = 0; I < N; ++I) {
         if (L.Role[Le * TME + I] != 2)
            continue;
#pragma unroll
         for (int J = 1; J < N; ++J) {
            const int Kk     = (I + J) % N;
            const T NormVort = (QRe[I] + QFe[I] + QRe[Kk] + QFe[Kk]) * 0.5;
            Acc[I] += L.Wt[(Le * TME + I) * TM1 + J - 1] * Flux[Kk] * Uj[Kk] * NormVort;
         }
      }
      // ---- remaining terms; this cell is CellsOnEdge(e,1) of every edge it finishes ----
      T Rv[N], R2[N];
#pragma unroll
      for (int J = 0; J < N; ++J) {
         Rv[J] = ldo<T>(RelVort, OffV[J]);
         R2[J] = ldo<T>(Del2RelVort, OffV[J]);
      }
      const T KEs = ldo<T>(KE, OffS), DivS = ldo<T>(Div, OffS), D2S = ldo<T>(Del2Div, OffS);
      const T Ssh1 = Hs - L.BDs[Le];
#pragma unroll
      for (int I = 0; I < N; ++I) {
         const int Li = Le * TME + I;
         if (L.Role[Li] != 2)
            continue;
         const int Im     = (I + N - 1) % N;
         const Real InvDc = L.InvDc[Li], InvDvS = L.InvDvS[Li];
         StagePre<T> PreU{};
         if (EPI)
            PreU = stagePre<T, true>(E, OffE[I]);
         T TendV = splat<T>(0.0);
         TendV += Acc[I]; // EdgeMask is 1 on a regular edge
         TendV -= (KEs - ldo<T>(KE, OffN[I])) * InvDc;
         const T Ssh0 = Hn[I] - L.BDn[Li];
         TendV -= Grav * (Ssh1 - Ssh0) * InvDc;
         {
            const T Del2U = ((DivS - ldo<T>(Div, OffN[I])) * InvDc - (Rv[I] - Rv[Im]) * InvDvS);
            TendV += L.C2[Li] * Del2U;
         }
         {
            const T Del2U = (P.DivFactor * (D2S - ldo<T>(Del2Div, OffN[I])) * InvDc - (R2[I] - R2[Im]) * InvDvS);
            TendV -= L.C4[Li] * Del2U;
         }
         if (!EPI || E.StoreTend)
            stnt<T>(Tend, OffE[I], TendV);
         if (EPI)
            stageApply<T, true>(E, OffE[I], TendV, Uj[I], PreU);
      }
   }
};

// CellPVFinalBody<TME, TME> and the default-term FusedCell3Body in one thread: the L3 work of a cell with one gather
// of h and u (32 B per cell-level less than the paired launch of the two kernels; same expressions, so same bits).
// Plain RHS only: with the stage updates in the epilogues the paired launch is the faster one.
template <int TME, int NR = TME, int FL = 3> struct CellPVFinalTracerBody {
   __device__ __forceinline__ int cellOf(int I) const {
      if constexpr ((FL & 1) != 0)
         return List ? List[I] : I;
      else
         return I;
   }
   static constexpr int MinWaves = OMEGA_PVF_MINW;
   static constexpr int TM1      = TME - 1;
   MeshView M;
   int K, NT;
   TendParams P;
   const Real *H, *U, *RelVortV, *InvThickV, *Partial;
   const Real *RelVort, *KE, *Div, *Del2Div, *Del2RelVort;
   Real *Tend;
   const Real *Tr, *Del2Tr;
   Real *TrTend;
   const int *List = nullptr;
   struct Lds {
      Real *Wt, *InvDc, *InvDvS, *C2, *C4, *BDn, *CellS, *FV, *MDvS, *Df2, *Df4; // CellS[2 Le] = BottomDepth, [2 Le + 1] = 1/Area
      int *Edge, *NbrF, *Ring, *Role, *N;
   };
   __host__ __device__ size_t ldsBytes(int Tile) const {
      return ldsRound8(sizeof(Real) * Tile * TME * TM1) + ldsRound8(sizeof(Real) * Tile * TME) * 9 +
             ldsRound8(sizeof(Real) * Tile) * 2 + ldsRound8(sizeof(int) * Tile * TME) * 4 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.Wt     = C.take<Real>(Tile * TME * TM1);
      L.InvDc  = C.take<Real>(Tile * TME);
      L.InvDvS = C.take<Real>(Tile * TME);
      L.C2     = C.take<Real>(Tile * TME);
      L.C4     = C.take<Real>(Tile * TME);
      L.BDn    = C.take<Real>(Tile * TME);
      L.FV     = C.take<Real>(Tile * TME);
      L.MDvS   = C.take<Real>(Tile * TME);
      L.Df2    = C.take<Real>(Tile * TME);
      L.Df4    = C.take<Real>(Tile * TME);
      L.CellS  = C.take<Real>(Tile * 2);
      L.Edge   = C.take<int>(Tile * TME);
      L.NbrF   = C.take<int>(Tile * TME);
      L.Ring   = C.take<int>(Tile * TME);
      L.Role   = C.take<int>(Tile * TME);
      L.N      = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt * TME * TM1; I += NThr) {
         const int Le = I / (TME * TM1);
         const int C  = cellOf(First + Le);
         L.Wt[I]      = M.PVWeightOnCell[(size_t)C * TME * TM1 + (I - Le * TME * TM1)];
      }
      for (int I = Tid; I < Cnt * TME; I += NThr) {
         const int Le    = I / TME;
         const int C     = cellOf(First + Le);
         const size_t G  = (size_t)C * TME + (I - Le * TME);
         const int Ed    = M.EdgesOnCell[G];
         const int F     = M.NbrFlagOnCell[G];
         const Real Mask = M.EdgeMask1D[Ed];
         L.Edge[I]       = Ed;
         L.NbrF[I]       = F;
         L.Ring[I]       = M.RingVertOnCell[G];
         L.FV[I]         = M.FVertex[M.RingVertOnCell[G]];
         L.Role[I]       = M.PVRoleOnCell[G];
         L.InvDc[I]      = M.InvDcEdge[Ed];
         L.InvDvS[I]     = M.RingSignOnCell[G] * M.InvDvEdge[Ed];
         L.C2[I]         = Mask * P.ViscDel2 * M.MeshScalingDel2[Ed];
         L.C4[I]         = Mask * P.ViscDel4 * M.MeshScalingDel4[Ed];
         L.BDn[I]        = M.BottomDepth[F & 0x3fffffff];
         L.MDvS[I]       = M.MaskDvSignOnCell[G];
         L.Df2[I]        = M.Diff2CoefSOnCell[G];
         L.Df4[I]        = M.Diff4CoefSOnCell[G];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         const int C = cellOf(First + I);
         L.CellS[2 * I]     = M.BottomDepth[C];
         L.N[I]             = M.NEdgesOnCell[C];
         L.CellS[2 * I + 1] = M.InvAreaCell[C];
      }
   }
   /// what the velocity part gathers and the tracer loop goes on with
   template <class T> struct RingVals {
      unsigned OffS, OffN[TME];
      T Hs, Hn[TME], Uj[TME];
   };
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      if constexpr ((FL & 2) != 0) {
         if (L.N[Le] > TME)
            return; // a cell wider than these tables: it has its own (list) launches on the wide tables
      }
      RingVals<T> R;
      velPart<T>(L, Le, cellOf(IElem), Kv, R);
      tracerLoopDirect<T>(L, Le, R);
   }
   /// CellPVFinalBody<TME, NR>: finishes the edges of which this cell is the second cell
   template <class T> __device__ __forceinline__ void velPart(const Lds &L, int Le, int ICell, int Kv, RingVals<T> &R) const {
      const Real Grav = 9.80665; // TendencyTerms.h:176
      unsigned OffE[TME];
      unsigned(&OffN)[TME] = R.OffN;
      T(&Uj)[TME] = R.Uj;
      T(&Hn)[TME] = R.Hn;
      const unsigned OffS = R.OffS = rowOff<T>(ICell, K, Kv);
      const T Hs = R.Hs   = ldo<T>(H, OffS);
#pragma unroll
      for (int J = 0; J < TME; ++J) {
         OffE[J] = rowOff<T>(L.Edge[Le * TME + J], K, Kv);
         OffN[J] = rowOff<T>(L.NbrF[Le * TME + J] & 0x3fffffff, K, Kv);
         Uj[J]   = ldo<T>(U, OffE[J]);
         Hn[J]   = ldo<T>(H, OffN[J]);
      }
      bool Any = L.N[Le] == NR;
      if (Any) {
         Any = false;
#pragma unroll
         for (int J = 0; J < NR; ++J)
            Any |= L.Role[Le * TME + J] == 2;
      }
      if (Any) { // ---- CellPVFinalBody<TME, NR> ----
         constexpr int N = NR;
         unsigned OffV[N];
         T Flux[N], QRe[N], QFe[N];
         {
            T QR[N], QF[N];
#pragma unroll
            for (int J = 0; J < N; ++J) {
               OffV[J]    = rowOff<T>(L.Ring[Le * TME + J], K, Kv);
               const T Iv = ldo<T>(InvThickV, OffV[J]);
               QR[J]      = ldo<T>(RelVortV, OffV[J]) * Iv;
               QF[J]      = L.FV[Le * TME + J] * Iv;
            }
#pragma unroll
            for (int J = 0; J < N; ++J) {
               Flux[J]      = 0.5 * (Hs + Hn[J]);
               const int Jm = (J + N - 1) % N;
               QRe[J]       = 0.5 * (QR[Jm] + QR[J]);
               QFe[J]       = 0.5 * (QF[Jm] + QF[J]);
            }
         }
         // the running sums of the edges this cell finishes: asked for together, ahead of the per-edge blocks (a load
         // inside a block is one more dependent round trip per block); switched off for the other slots
         T Acc[N];
#pragma unroll
         for (int I = 0; I < N; ++I)
            Acc[I] = ldntIf<T>(L.Role[Le * TME + I] == 2, Partial, OffE[I]);
#pragma unroll
         for (int I = 0; I < N; ++I) {
            if (L.Role[Le * TME + I] != 2)
               continue;
#pragma unroll
            for (int J = 1; J < N; ++J) {
               const int Kk     = (I + J) % N;
               const T NormVort = (QRe[I] + QFe[I] + QRe[Kk] + QFe[Kk]) * 0.5;
               Acc[I] += L.Wt[(Le * TME + I) * TM1 + J - 1] * Flux[Kk] * Uj[Kk] * NormVort;
            }
         }
         T Rv[N], R2[N];
#pragma unroll
         for (int J = 0; J < N; ++J) {
            Rv[J] = ldo<T>(RelVort, OffV[J]);
            R2[J] = ldo<T>(Del2RelVort, OffV[J]);
         }
         const T KEs = ldo<T>(KE, OffS), DivS = ldo<T>(Div, OffS), D2S = ldo<T>(Del2Div, OffS);
         const T Ssh1 = Hs - L.CellS[2 * Le];
#pragma unroll
         for (int I = 0; I < N; ++I) {
            const int Li = Le * TME + I;
            if (L.Role[Li] != 2)
               continue;
            const int Im     = (I + N - 1) % N;
            const Real InvDc = L.InvDc[Li], InvDvS = L.InvDvS[Li];
            const T KEnI = ldo<T>(KE, OffN[I]), DivNI = ldo<T>(Div, OffN[I]), D2NI = ldo<T>(Del2Div, OffN[I]);
            T TendV = splat<T>(0.0);
            TendV += Acc[I];
            TendV -= (KEs - KEnI) * InvDc;
            const T Ssh0 = Hn[I] - L.BDn[Li];
            TendV -= Grav * (Ssh1 - Ssh0) * InvDc;
            {
               const T Del2U = ((DivS - DivNI) * InvDc - (Rv[I] - Rv[Im]) * InvDvS);
               TendV += L.C2[Li] * Del2U;
            }
            {
               const T Del2U = (P.DivFactor * (D2S - D2NI) * InvDc - (R2[I] - R2[Im]) * InvDvS);
               TendV -= L.C4[Li] * Del2U;
            }
            stnt<T>(Tend, OffE[I], TendV);
         }
      }
   }
   /// FusedCell3Body<TME, true>: every tracer's neighbour values gathered by the thread
   template <class T> __device__ __forceinline__ void tracerLoopDirect(const Lds &L, int Le, const RingVals<T> &R) const {
      const unsigned OffS = R.OffS;
      const unsigned(&OffN)[TME] = R.OffN;
      const T Hs = R.Hs;
      const T(&Hn)[TME] = R.Hn;
      const T(&Uj)[TME] = R.Uj;
      const Real InvA      = L.CellS[2 * Le + 1];
      const size_t CStride = (size_t)M.NCellsSize * K;
#ifndef OMEGA_L3_TRUNROLL
#define OMEGA_L3_TRUNROLL 1
#endif
      constexpr int TU = TME <= 6 ? OMEGA_L3_TRUNROLL : 1; // tracers per trip (see FusedCellL1PVBody); registers
#pragma nounroll
      for (int Lt = 0; Lt < NT; Lt += TU) {
         loopFence();
         T Tn[TU][TME], Dn[TU][TME], Ts[TU], Ds[TU];
#pragma unroll
         for (int Q = 0; Q < TU; ++Q) {
            const bool Valid = TU == 1 || Lt + Q < NT;
            const Real *TrL  = uniformPtr(Tr + (Valid ? Lt + Q : Lt) * CStride);
            const Real *D2L  = uniformPtr(Del2Tr + (Valid ? Lt + Q : Lt) * CStride);
#pragma unroll
            for (int J = 0; J < TME; ++J) {
               Tn[Q][J] = ldoIf<T>(Valid, TrL, OffN[J]);
               Dn[Q][J] = ldoIf<T>(Valid, D2L, OffN[J]);
            }
            Ts[Q] = ldoIf<T>(Valid, TrL, OffS);
            Ds[Q] = ldoIf<T>(Valid, D2L, OffS);
         }
#pragma unroll
         for (int Q = 0; Q < TU; ++Q) {
            const bool Valid = TU == 1 || Lt + Q < NT;
            T HAdvTmp = splat<T>(0.0), DiffTmp = splat<T>(0.0), HypTmp = splat<T>(0.0);
            const T HsTs = Hs * Ts[Q];
#pragma unroll
            for (int J = 0; J < TME; ++J) {
               const int I  = Le * TME + J;
               const T HTr  = 0.5 * (HsTs + Hn[J] * Tn[Q][J]);
               HAdvTmp -= L.MDvS[I] * HTr * Uj[J] * InvA;
               const T Mean = 0.5 * (Hs + Hn[J]);
               DiffTmp -= L.Df2[I] * Mean * (Tn[Q][J] - Ts[Q]);
               HypTmp -= L.Df4[I] * (Dn[Q][J] - Ds[Q]);
            }
            T TendV = splat<T>(0.0);
            TendV -= HAdvTmp;
            TendV += P.EddyDiff2 * DiffTmp * InvA;
            TendV -= P.EddyDiff4 * HypTmp * InvA;
            stntIf<T>(Valid, uniformPtr(TrTend + (Valid ? Lt + Q : Lt) * CStride), OffS, TendV);
         }
      }
   }
};


// ---------------------------------------------------------------------------------------
// CellPVFinalTracerBody with the tracer loop's neighbour values staged through LDS once per workgroup (tile patches,
// HorzMesh.h).  Per (tile, level chunk, tracer) the workgroup moves the chunk's 128 bytes of every row of the tile's patch
// -- Tr and Del2Tr: 2 x PatchNP rows -- from the buffers straight into LDS (`buffer_load_dwordx4 ... lds`: 16 bytes per
// lane, 8 rows per instruction, no registers), double-buffered: tracer t+1 is in flight while tracer t is computed, one
// workgroup barrier per tracer.  A thread then reads its 2 x 7 values from LDS.  Against the per-thread gathers: each
// row is asked for once instead of by up to seven threads, the loop's loads never wait for registers, and its compute
// phase always has the next tracer's requests outstanding.  Same expressions in the same order: same bits.
// The transfers are inline assembly on purpose: the compiler makes every LDS read wait for all LDS-DMA it knows of, which
// would serialise the pipeline; hidden from it, the waits are placed by hand (memory operations of a wave complete in
// order, so the compiler's own counts stay safe -- they can only wait for more than they need).
template <int TME, int NR = TME, int FL = 3> struct CellPVFinalTracerPatchBody : CellPVFinalTracerBody<TME, NR, FL> {
   using Base = CellPVFinalTracerBody<TME, NR, FL>;
   static constexpr bool Cooperative = true; // the tile kernels call computeTile() with every thread of the workgroup
   const I4 *PRows, *PIdx, *POK; // the mesh's patch tables for this launch's tile size
   int NP;                       // rows per patch (multiple of 8)
   int PatchTile;                // the tile size these tables were built for (checked against the launch's)
   int NWv = 4;                  // wavefronts per workgroup (KernelCommon.h: setWaves)
   struct Lds : Base::Lds {
      int *PRow;
      unsigned char *PIdxB, *Buf; // Buf: [2 buffers][2 arrays][NP rows][128 bytes]
      int *OKp;
   };
   size_t ldsBytes(int Tile) const {
      return Base::ldsBytes(Tile) + ldsRound8(sizeof(int) * NP) + ldsRound8((size_t)Tile * 8) + 16 + 8 + (size_t)4 * NP * 128;
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      Lds L;
      static_cast<typename Base::Lds &>(L) = Base::carve(Ptr, Tile);
      LdsCarver C{Ptr + Base::ldsBytes(Tile)};
      L.PRow  = C.take<int>(NP);
      L.PIdxB = C.take<unsigned char>(Tile * 8);
      L.OKp   = C.take<int>(2);
      L.Buf   = C.P + ((16u - ((unsigned)(C.P - Ptr) & 15u)) & 15u); // (pointer arithmetic only: stays an LDS address)
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      Base::stage(L, First, Cnt, Tid, NThr);
      const int Tile = blockDim.y, Tl = First / Tile;
      if (Tile == PatchTile) {
         for (int I = Tid; I < NP; I += NThr)
            L.PRow[I] = PRows[(size_t)Tl * NP + I];
         const unsigned char *Src = reinterpret_cast<const unsigned char *>(PIdx) + (size_t)First * 8;
         for (int I = Tid; I < Cnt * 8; I += NThr)
            L.PIdxB[I] = Src[I];
      }
      if (Tid == 0) // (a launch whose geometry is not the tables': the per-thread gathers)
         L.OKp[0] = (Tile == PatchTile && blockDim.x == 8) ? POK[Tl] : 0;
   }
   /// one 16-byte-per-lane transfer buffer -> LDS; LdsAddr = LDS byte address of the wavefront's 1 KiB destination
   __device__ __forceinline__ static void dma16(const Real *Plane, unsigned ByteOff, unsigned LdsAddr) {
      const unsigned long long V = reinterpret_cast<unsigned long long>(Plane);
      BufV4 Rs;
      Rs.x = __builtin_amdgcn_readfirstlane((unsigned)V);
      Rs.y = __builtin_amdgcn_readfirstlane((unsigned)(V >> 32) & 0xffffu);
      Rs.z = BufOOB;
      Rs.w = 0x00020000u;
      const unsigned A = __builtin_amdgcn_readfirstlane(LdsAddr); // (wave-uniform by construction; the compiler may hold it in a VGPR)
      __asm__ volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds"
                       :
                       : "s"(A), "v"(ByteOff), "s"(Rs)
                       : "memory", "m0"); // (M0 is written: the compiler must not keep a value of its own in it across this)
   }
   /// this wavefront's share of tracer Lt's rows into buffer (It & 1)
   template <class T> __device__ __forceinline__ void fetchTracer(const Lds &L, int Lt, unsigned It, int Kv, bool KvOK) const {
      const unsigned Tid  = threadIdx.y * blockDim.x + threadIdx.x;
      const unsigned Wv   = __builtin_amdgcn_readfirstlane(Tid >> 6);
      const unsigned Lane = Tid & 63u;
      const size_t CStride = (size_t)this->M.NCellsSize * this->K;
      const Real *TrL = uniformPtr(this->Tr + Lt * CStride), *D2L = uniformPtr(this->Del2Tr + Lt * CStride);
      const unsigned BufBase =
          __builtin_amdgcn_readfirstlane((unsigned)reinterpret_cast<size_t>(L.Buf)) + (It & 1u) * 2u * (unsigned)NP * 128u;
      for (int G = (int)Wv; G * 8 < NP; G += NWv) { // 8 rows (1 KiB) per transfer
         const int Row      = L.PRow[G * 8 + (int)(Lane >> 3)];
         const unsigned Off = (Row >= 0 && KvOK) ? rowOff<T>(Row, this->K, Kv) : BufOOB;
         dma16(TrL, Off, BufBase + (unsigned)G * 1024u);
         dma16(D2L, Off, BufBase + (unsigned)NP * 128u + (unsigned)G * 1024u);
      }
   }
   template <class T> __device__ void computeTile(const Lds &L, int First, int Cnt, int C0, int CS, int KV) const {
      if constexpr (sizeof(T) != 16) { // (the launcher only takes this body with 16-byte accesses; never run)
         for (int Le = threadIdx.y; Le < Cnt; Le += blockDim.y)
            for (int Kv = C0 * blockDim.x + threadIdx.x; Kv < KV; Kv += blockDim.x * CS)
               Base::template compute<T>(L, Le, First + Le, Kv);
         return;
      }
      const int Le    = threadIdx.y;
      const bool Mine = Le < Cnt && !(((FL & 2) != 0) && L.N[Le < Cnt ? Le : 0] > TME);
      const int ICell = this->cellOf(First + (Le < Cnt ? Le : 0));
      const bool Patch = L.OKp[0] != 0; // (workgroup-uniform)
      unsigned It = 0;                  // transfers issued so far by this workgroup: buffer = It & 1
      for (int Kc = C0; Kc * (int)blockDim.x < KV; Kc += CS) { // (the same trip count for every thread)
         chunkFence<CellPVFinalTracerPatchBody>();
         const int Kv    = Kc * blockDim.x + threadIdx.x;
         const bool KvOK = Kv < KV;
         const bool Act  = Mine && KvOK;
         const int LeT = Le < Cnt ? Le : 0; // this thread's tile-local cell (0 for a thread without one)
         // the first tracer's rows travel while the velocity part runs (the buffer was last read two tracers ago, and
         // every wave has passed the barrier of the tracer in between)
         if (Patch)
            fetchTracer<T>(L, 0, It, Kv, KvOK);
         typename Base::template RingVals<T> R;
         R.OffS = BufOOB, R.Hs = splat<T>(0.0);
#pragma unroll
         for (int J = 0; J < TME; ++J)
            R.OffN[J] = BufOOB, R.Hn[J] = splat<T>(0.0), R.Uj[J] = splat<T>(0.0);
         if (Act)
            Base::template velPart<T>(L, LeT, ICell, Kv, R);
         if (!Patch) { // a tile whose patch does not fit (one code path for the velocity part either way): per-thread gathers
            if (Act)
               Base::template tracerLoopDirect<T>(L, LeT, R);
            continue;
         }
         const unsigned OffS = Act ? R.OffS : BufOOB; // (inactive threads run the loop for its barriers; they store nothing)
         const Real InvA      = L.CellS[2 * LeT + 1];
         const size_t CStride = (size_t)this->M.NCellsSize * this->K;
         const unsigned char *PI = L.PIdxB + LeT * 8;
         unsigned PO[TME + 1]; // byte offsets of this thread's 7 + 1 values inside a buffer plane
#pragma unroll
         for (int J = 0; J < TME; ++J)
            PO[J] = (unsigned)PI[J] * 128u + threadIdx.x * 16u;
         PO[TME] = (unsigned)PI[7] * 128u + threadIdx.x * 16u;
#pragma nounroll
         for (int Lt = 0; Lt < this->NT; ++Lt, ++It) {
            loopFence();
            // tracer Lt's rows have landed: everything this wave asked for before its most recent store
            if (Lt == 0)
               __asm__ volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
               __asm__ volatile("s_waitcnt vmcnt(1)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __asm__ volatile("" ::: "memory");
            if (Lt + 1 < this->NT)
               fetchTracer<T>(L, Lt + 1, It + 1, Kv, KvOK);
            const unsigned char *BT = L.Buf + (size_t)(It & 1u) * 2u * NP * 128u, *BD = BT + (size_t)NP * 128u;
            T Tn[TME], Dn[TME];
#pragma unroll
            for (int J = 0; J < TME; ++J) {
               Tn[J] = *reinterpret_cast<const T *>(BT + PO[J]);
               Dn[J] = *reinterpret_cast<const T *>(BD + PO[J]);
            }
            const T Ts = *reinterpret_cast<const T *>(BT + PO[TME]), Ds = *reinterpret_cast<const T *>(BD + PO[TME]);
            T HAdvTmp = splat<T>(0.0), DiffTmp = splat<T>(0.0), HypTmp = splat<T>(0.0);
            const T HsTs = R.Hs * Ts;
#pragma unroll
            for (int J = 0; J < TME; ++J) {
               const int I  = LeT * TME + J;
               const T HTr  = 0.5 * (HsTs + R.Hn[J] * Tn[J]);
               HAdvTmp -= L.MDvS[I] * HTr * R.Uj[J] * InvA;
               const T Mean = 0.5 * (R.Hs + R.Hn[J]);
               DiffTmp -= L.Df2[I] * Mean * (Tn[J] - Ts);
               HypTmp -= L.Df4[I] * (Dn[J] - Ds);
            }
            T TendV = splat<T>(0.0);
            TendV -= HAdvTmp;
            TendV += this->P.EddyDiff2 * DiffTmp * InvA;
            TendV -= this->P.EddyDiff4 * HypTmp * InvA;
            stnt<T>(uniformPtr(this->TrTend + Lt * CStride), OffS, TendV); // (exactly one store per tracer: the wait above counts on it)
         }
      }
   }
};

// L3 edge pass after the cell-centric PV sums: the remaining velocity terms for regular edges,
// with the finished PV sum read from `Partial`.
template <bool Fast> struct EdgeFinalBody {
   MeshView M;
   int K;
   TendParams P;
   const Real *H, *U, *Partial;
   const Real *RelVort, *KE, *Div, *Del2Div, *Del2RelVort, *NormalStress;
   Real *Tend;
   int KLog = 0; ///< number of levels (K is the row pitch): set by launchTile
   struct Lds {
      Real *InvDc, *InvDv, *Mask, *MaskGrav, *C2, *C4, *BD0, *BD1;
      int *C0, *C1, *V0, *V1, *Reg;
   };
   size_t ldsBytes(int Tile) const { return ldsRound8(sizeof(Real) * Tile) * 8 + ldsRound8(sizeof(int) * Tile) * 5; }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.InvDc    = C.take<Real>(Tile);
      L.InvDv    = C.take<Real>(Tile);
      L.Mask     = C.take<Real>(Tile);
      L.MaskGrav = C.take<Real>(Tile);
      L.C2       = C.take<Real>(Tile);
      L.C4       = C.take<Real>(Tile);
      L.BD0      = C.take<Real>(Tile);
      L.BD1      = C.take<Real>(Tile);
      L.C0       = C.take<int>(Tile);
      L.C1       = C.take<int>(Tile);
      L.V0       = C.take<int>(Tile);
      L.V1       = C.take<int>(Tile);
      L.Reg      = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      const Real Grav = 9.80665; // TendencyTerms.h:176
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
         L.Reg[I]      = M.EdgeRegular[E];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IEdge, int Kv) const {
      if (!L.Reg[Le])
         return; // irregular edges are written by the edge-centric kernel
      const bool PVOn = Fast ? true : (P.PVTendencyEnable != 0), KEOn = Fast ? true : (P.KETendencyEnable != 0);
      const bool SSHOn = Fast ? true : (P.SSHTendencyEnable != 0), D2On = Fast ? true : (P.VelDiffTendencyEnable != 0);
      const bool D4On   = Fast ? true : (P.VelHyperDiffTendencyEnable != 0);
      const bool WindOn = Fast ? false : (P.WindForcingTendencyEnable != 0);
      const bool DragOn = Fast ? false : (P.BottomDragTendencyEnable != 0);
      const unsigned OffC0 = rowOff<T>(L.C0[Le], K, Kv), OffC1 = rowOff<T>(L.C1[Le], K, Kv);
      const unsigned OffV0 = rowOff<T>(L.V0[Le], K, Kv), OffV1 = rowOff<T>(L.V1[Le], K, Kv);
      const unsigned OffE  = rowOff<T>(IEdge, K, Kv);
      const Real InvDc = L.InvDc[Le], InvDv = L.InvDv[Le];
      const T H0 = ldo<T>(H, OffC0), H1 = ldo<T>(H, OffC1);
      T TendV = splat<T>(0.0);
      if (PVOn)
         TendV += L.Mask[Le] * ldo<T>(Partial, OffE);
      if (KEOn)
         TendV -= L.Mask[Le] * (ldo<T>(KE, OffC1) - ldo<T>(KE, OffC0)) * InvDc;
      if (SSHOn) {
         const T Ssh0 = H0 - L.BD0[Le], Ssh1 = H1 - L.BD1[Le];
         TendV -= L.MaskGrav[Le] * (Ssh1 - Ssh0) * InvDc;
      }
      if (D2On) {
         const T Del2U = ((ldo<T>(Div, OffC1) - ldo<T>(Div, OffC0)) * InvDc -
                          (ldo<T>(RelVort, OffV1) - ldo<T>(RelVort, OffV0)) * InvDv);
         TendV += L.C2[Le] * Del2U;
      }
      if (D4On) {
         const T Del2U = (P.DivFactor * (ldo<T>(Del2Div, OffC1) - ldo<T>(Del2Div, OffC0)) * InvDc -
                          (ldo<T>(Del2RelVort, OffV1) - ldo<T>(Del2RelVort, OffV0)) * InvDv);
         TendV -= L.C4[Le] * Del2U;
      }
      constexpr int W = VecW<T>::W;
      if (WindOn && Kv == 0) {
         const Real HMean0       = 0.5 * (getc(H0, 0) + getc(H1, 0));
         const Real InvThickEdge = 1. / HMean0;
         setc(TendV, 0, getc(TendV, 0) + L.Mask[Le] * InvThickEdge * NormalStress[IEdge] / P.Density0);
      }
      if (DragOn && (Kv + 1) * W >= KLog) {
         const int KBot          = KLog - 1;
         const int Comp          = KBot - Kv * W;
         const Real VelNormEdge  = sqrt(KE[(size_t)L.C0[Le] * K + KBot] + KE[(size_t)L.C1[Le] * K + KBot]);
         const Real HMeanB       = 0.5 * (getc(H0, Comp) + getc(H1, Comp));
         const Real InvThickEdge = 1. / HMeanB;
         setc(TendV, Comp,
              getc(TendV, Comp) - L.Mask[Le] * P.BottomDragCoeff * VelNormEdge * InvThickEdge * U[(size_t)IEdge * K + KBot]);
      }
      stnt<T>(Tend, OffE, TendV);
   }
};

// ---------------------------------------------------------------------------------------
// L3 cell pass: tracer tendencies (TendencyTerms.h:349-480) with HTracersEdge
// (TracerAuxVars.h:25-59) and MeanLayerThickEdge rebuilt inline; tracer loop inside.
template <int TME, bool Fast, bool EPI = false, int FL = 3> struct FusedCell3Body {
   static constexpr int MinWaves = OMEGA_C3_MINW;
   static constexpr int MaxW     = OMEGA_CELL_MAXW;
   MeshView M;
   int K, NT;
   TendParams P;
   const Real *H, *U, *Tr, *Del2Tr;
   Real *Tend;
   StageEpi E{};              // tracer stage update (EPI)
   const int *List = nullptr; // optional cell list (band / interior launches of an overlapped exchange; wide cells)
   struct Lds {
      Real *MDvS, *Df2, *Df4, *InvA;
      int *Edge, *NbrF, *N;
   };
   size_t ldsBytes(int Tile) const {
      return ldsRound8(sizeof(Real) * Tile * TME) * 3 + ldsRound8(sizeof(Real) * Tile) +
             ldsRound8(sizeof(int) * Tile * TME) * 2 + ldsRound8(sizeof(int) * Tile);
   }
   __device__ Lds carve(unsigned char *Ptr, int Tile) const {
      LdsCarver C{Ptr};
      Lds L;
      L.MDvS = C.take<Real>(Tile * TME);
      L.Df2  = C.take<Real>(Tile * TME);
      L.Df4  = C.take<Real>(Tile * TME);
      L.InvA = C.take<Real>(Tile);
      L.Edge = C.take<int>(Tile * TME);
      L.NbrF = C.take<int>(Tile * TME);
      L.N    = C.take<int>(Tile);
      return L;
   }
   __device__ void stage(const Lds &L, int First, int Cnt, int Tid, int NThr) const {
      for (int I = Tid; I < Cnt * TME; I += NThr) {
         const int Le   = I / TME;
         const int C    = List ? List[First + Le] : First + Le;
         const size_t G = (size_t)C * TME + (I - Le * TME);
         L.MDvS[I]      = M.MaskDvSignOnCell[G];
         L.Df2[I]       = Fast ? M.Diff2CoefSOnCell[G] : M.Diff2CoefOnCell[G];
         L.Df4[I]       = Fast ? M.Diff4CoefSOnCell[G] : M.Diff4CoefOnCell[G];
         L.Edge[I]      = M.EdgesOnCell[G];
         L.NbrF[I]      = M.NbrFlagOnCell[G];
      }
      for (int I = Tid; I < Cnt; I += NThr) {
         const int C = List ? List[First + I] : First + I;
         L.InvA[I]   = M.InvAreaCell[C];
         L.N[I]      = M.NEdgesOnCell[C];
      }
   }
   template <class T> __device__ void compute(const Lds &L, int Le, int IElem, int Kv) const {
      if constexpr ((FL & 2) != 0) {
         if (L.N[Le] > TME)
            return; // a cell wider than these tables: it has its own (list) launch on the wide tables
      }
      const int ICell     = List ? List[IElem] : IElem;
      const bool TrUpwind = Fast ? false : (P.FluxTracerUpwind != 0);
      const bool AdvOn = Fast ? true : (P.TracerHorzAdvTendencyEnable != 0);
      const bool DiffOn = Fast ? true : (P.TracerDiffTendencyEnable != 0);
      const bool HypOn  = Fast ? true : (P.TracerHyperDiffTendencyEnable != 0);
      const Real InvA     = L.InvA[Le];
      const unsigned OffS = rowOff<T>(ICell, K, Kv);
      unsigned OffN[TME];
      bool IsC0[TME];
      T UJ[TME], Hn[TME];
#pragma unroll
      for (int J = 0; J < TME; ++J) {
         const int F = L.NbrF[Le * TME + J];
         OffN[J]     = rowOff<T>(F & 0x3fffffff, K, Kv);
         IsC0[J]     = (F >> 30) != 0;
         UJ[J]       = ldo<T>(U, rowOff<T>(L.Edge[Le * TME + J], K, Kv));
         Hn[J]       = ldo<T>(H, OffN[J]);
      }
      const T Hs = ldo<T>(H, OffS);
      const size_t CStride = (size_t)M.NCellsSize * K;
      // stage update (EPI): thicknesses this cell's tracer update divides / multiplies by
      T EpCurH = Hs, EpDivH = Hs;
      if (EPI) { // (both asked for with the gathers above; the first is switched off where the stage does not read it)
         const T LCur = ldoIf<T>(!E.Last && !E.First, E.CurH, OffS);
         EpDivH       = ldo<T>(E.Last ? E.NextH : E.ProvH, OffS);
         EpCurH       = (!E.Last && !E.First) ? LCur : Hs;
      }
#pragma nounroll
      for (int Lt = 0; Lt < NT; ++Lt) {
         loopFence();
         const Real *TrL = uniformPtr(Tr + Lt * CStride);
         const Real *D2L = uniformPtr(Del2Tr + Lt * CStride);
         T Tn[TME], Dn[TME];
#pragma unroll
         for (int J = 0; J < TME; ++J) {
            Tn[J] = ldo<T>(TrL, OffN[J]);
            Dn[J] = HypOn ? ldo<T>(D2L, OffN[J]) : splat<T>(0.0);
         }
         const T Ts = ldo<T>(TrL, OffS);
         const T Ds = HypOn ? ldo<T>(D2L, OffS) : splat<T>(0.0);
         // stage update operands (EPI), asked for with the gathers instead of after the arithmetic; switched off
         // through the offset in the stages that do not read them
         Real *NextL = nullptr;
         T NextOld = Ts, CurOld = Ts;
         if (EPI) {
            NextL   = uniformPtr(E.Next + Lt * CStride);
            NextOld = ldntIf<T>(!E.First, NextL, OffS);
            CurOld  = ldntIf<T>(!E.First && !E.Last, uniformPtr(E.Cur + Lt * CStride), OffS);
         }
         T HAdvTmp = splat<T>(0.0), DiffTmp = splat<T>(0.0), HypTmp = splat<T>(0.0);
         if (Fast) {
            // center fluxes: h(c0)*tr(c0) + h(c1)*tr(c1) and h(c0)+h(c1) do not depend on which
            // of the two cells is "this" one (a+b == b+a), and the staged diffusion coefficients
            // carry the orientation of (T1-T0) = +-(Tn-Ts)
            const T HsTs = Hs * Ts;
#pragma unroll
            for (int J = 0; J < TME; ++J) {
               const int I  = Le * TME + J;
               const T HTr  = 0.5 * (HsTs + Hn[J] * Tn[J]);
               HAdvTmp -= L.MDvS[I] * HTr * UJ[J] * InvA;
               const T Mean = 0.5 * (Hs + Hn[J]);
               DiffTmp -= L.Df2[I] * Mean * (Tn[J] - Ts);
               HypTmp -= L.Df4[I] * (Dn[J] - Ds);
            }
         } else {
#pragma unroll
            for (int J = 0; J < TME; ++J) {
               const int I = Le * TME + J;
               const T H0 = pick(IsC0[J], Hs, Hn[J]), H1 = pick(IsC0[J], Hn[J], Hs);
               const T T0 = pick(IsC0[J], Ts, Tn[J]), T1 = pick(IsC0[J], Tn[J], Ts);
               if (AdvOn) {
                  const T HT0 = H0 * T0, HT1 = H1 * T1;
                  const T HTr = TrUpwind ? upwind(UJ[J], HT0, HT1) : T(0.5 * (HT0 + HT1));
                  HAdvTmp -= L.MDvS[I] * HTr * UJ[J] * InvA;
               }
               if (DiffOn) {
                  const T Mean = 0.5 * (H0 + H1);
                  DiffTmp -= L.Df2[I] * Mean * (T1 - T0);
               }
               if (HypOn) {
                  const T D0 = pick(IsC0[J], Ds, Dn[J]), D1 = pick(IsC0[J], Dn[J], Ds);
                  HypTmp -= L.Df4[I] * (D1 - D0);
               }
            }
         }
         T TendV = splat<T>(0.0);
         if (AdvOn)
            TendV -= HAdvTmp;
         if (DiffOn)
            TendV += P.EddyDiff2 * DiffTmp * InvA;
         if (HypOn)
            TendV -= P.EddyDiff4 * HypTmp * InvA;
         if (!EPI || E.StoreTend)
            stnt<T>(uniformPtr(Tend + Lt * CStride), OffS, TendV);
         if (EPI) {
            // weightTracers + accumulateTracersUpdate (+ finalizeTracersUpdate in the last stage)
            T Acc = E.First ? T(Ts * Hs) : NextOld;
            Acc   = Acc + E.CB * TendV;
            if (E.Last)
               Acc = Acc / EpDivH;
            stnt<T>(NextL, OffS, Acc);
            if (!E.Last) { // updateTracersByTend: (CurTr*CurH + CA*Tend) / ProvH
               const T CurT = E.First ? Ts : CurOld;
               stnt<T>(uniformPtr(E.Prov + Lt * CStride), OffS, (CurT * EpCurH + E.CA * TendV) / EpDivH);
            }
         }
      }
   }
};


/// Default.yml term set: every flag folds at compile time (see `Fast` above)
static inline bool isDefaultTermSet(const TendParams &P) {
   return P.ThicknessFluxTendencyEnable && P.PVTendencyEnable && P.KETendencyEnable && P.SSHTendencyEnable &&
          P.VelDiffTendencyEnable && P.VelHyperDiffTendencyEnable && !P.WindForcingTendencyEnable &&
          !P.BottomDragTendencyEnable && P.TracerHorzAdvTendencyEnable && P.TracerDiffTendencyEnable &&
          P.TracerHyperDiffTendencyEnable && !P.FluxThicknessUpwind && !P.FluxTracerUpwind;
}

/// ND = the valence the full sweeps of the cell-centric PV kernels are instantiated for: TME, or TME-1 when most
/// cells have one edge fewer than the widest (hexagons with a few heptagons).  NA = the other of the two; cells of
/// valence NA and TME-2 go through list launches.
/// HW: M is the NARROW view of a mesh with wider cells (*Wide the full-width one); without, the cell bodies of the sweeps
/// carry neither list selects nor width tests (FL above).
template <int TME, bool Fast, int ND = TME, bool HW = false>
void launchFusedT(const MeshView &M, int K, int NT, const TendParams &P, const AuxPtrs &A, Real *HTend,
                         Real *UTend, Real *TrTend, const Real *H, const Real *U, const Real *Tr, hipStream_t S,
                         hipEvent_t *Ev, Real *EdgeScratch, const StageUpdate *Stage, const MeshView *Wide = nullptr) {
   // Wide != nullptr: M is the mesh's NARROW view (cell tables TME wide) and *Wide the full-width one; the cells with
   // TW = TME+1 edges (Wide->WideCells: the heptagons of a hexagon mesh) are skipped by every sweep over M and run
   // through list launches of the TW-slot bodies on *Wide, level by level.
   constexpr int TW       = TME < 8 ? TME + 1 : TME;
   constexpr bool CanWide = HW && ND == TME && TME < 8;
   constexpr int FLS      = HW ? 2 : 0; // flags of a full sweep: no list; width test only next to wider cells
   constexpr int FLL      = HW ? 3 : 1; // ... of a body that may also run over band / interior lists
   const I4 NWide         = (CanWide && Wide) ? Wide->NWideCells : 0;
   constexpr int NA     = ND == TME ? TME - 1 : TME;
   const I4 NMain       = ND == TME ? M.NRingCellsM0 : M.NRingCellsM1; // cells of the sweeps' valence
   const I4 NOther      = ND == TME ? M.NRingCellsM1 : M.NRingCellsM0; // cells of valence NA (list launches)
   const I4 *OtherCells = ND == TME ? M.RingCellsM1 : M.RingCellsM0;
   // Runge-Kutta stage update folded into the tendency-producing kernels (Fast term set only;
   // launchFusedRHS has checked that this mesh takes the cell-centric PV path)
   [[maybe_unused]] StageEpi EH, EU, ET;
   if (Stage) {
      EH.CB = EU.CB = ET.CB = Stage->CB, EH.CA = EU.CA = ET.CA = Stage->CA;
      EH.First = EU.First = ET.First = Stage->First, EH.Last = EU.Last = ET.Last = Stage->Last;
      EH.StoreTend = EU.StoreTend = ET.StoreTend = Stage->StoreTend;
      EH.Next = Stage->NextH, EH.Cur = Stage->CurH, EH.Prov = Stage->ProvH;
      EU.Next = Stage->NextU, EU.Cur = Stage->CurU, EU.Prov = Stage->ProvU;
      ET.Next = Stage->NextTr, ET.Cur = Stage->CurTr, ET.Prov = Stage->ProvTr;
      ET.CurH = Stage->CurH, ET.ProvH = Stage->ProvH, ET.NextH = Stage->NextH;
   }
   auto Mark = [&](int I) {
      if (Ev)
         (void)hipEventRecord(Ev[I], S);
   };
   // L1: replaces AuxState:vertexAuxState1, cellAuxState1, edgeAuxState1/2 (flux thickness), cellAuxState4 (Del2Tracers),
   // Tend:thicknessFluxDiv and the cell-0 half of Tend:potientialVortHAdv
   Pacer::start("Tend:fused:L1[AuxState:vertexAuxState1,cellAuxState1,edgeAuxState2,cellAuxState4;Tend:thicknessFluxDiv]", 2);
   Mark(0);
   // the vertex kernel stores RelVort and 1/LayerThickVertex; the two normalised vorticities are rebuilt from
   // them where they are consumed; without the cell-centric tables the edge kernels read the reference's arrays
   const TuningOptions &Tn = tuning();
   // the band of an overlapped stage: without the halo cells whose results the exchange replaces (Kernels.h)
   const bool SendOnly =
       Tn.SendBand && Stage && Stage->AfterBand && Stage->HaloOutputsReplaced && !Stage->StoreTend && M.NBandSendCells > 0;
   const I4 *const BandList = SendOnly ? M.BandSendCells : M.BandCells;
   const int NBandList      = SendOnly ? M.NBandSendCells : M.NBandCells;
   // sweep lengths of a stage (Kernels.h: StageUpdate::NCellsL1 / NCellsVel / NCellsTr)
   auto SweepLen = [&](I4 Want) {
      return (Tn.ShrinkSweeps && Stage && !Stage->StoreTend && Want > 0 && Want < M.NCellsAll) ? Want : M.NCellsAll;
   };
   const int NSweepL1 = SweepLen(Stage ? Stage->NCellsL1 : 0), NSweepVel = SweepLen(Stage ? Stage->NCellsVel : 0),
             NSweepTr = SweepLen(Stage ? Stage->NCellsTr : 0);
   // the stream of the band launches (Kernels.h: StageUpdate::BandStream); forked from S at the first use
   bool BandForked = false;
   auto BandS      = [&]() -> hipStream_t {
      if (!(Stage && Stage->BandStream && Stage->BandReady && Tn.BandOnComm))
         return S;
      if (!BandForked) {
         HIP_CHECK(hipEventRecord(Stage->BandReady, S));
         HIP_CHECK(hipStreamWaitEvent(Stage->BandStream, Stage->BandReady, 0));
         BandForked = true;
      }
      return Stage->BandStream;
   };
   const bool CellCentric     = M.CellPVOK && EdgeScratch;
   // vertex pass and side-0 PV sums inside the L1 cell kernel (option MergeL1 = 0: the three separate kernels)
   const int MergeL1Env = Tn.MergeL1;
   const bool MergeL1 = CellCentric && M.CellL1OK && P.PVTendencyEnable && MergeL1Env != 0 &&
                        (Fast || TME <= 7); // (8 edge slots with run-time option flags would spill registers)
   FusedKernelNames[0]        = MergeL1 ? "" : "VortVertexBody";
   FusedKernelNames[1]        = MergeL1 ? "FusedCellL1PVBody" : "FusedCell1Body";
   if (!MergeL1)
      launchVertexAuxState1(Wide ? *Wide : M, K, A, H, U, S, /*StoreNorm*/ !CellCentric, /*StoreInv*/ CellCentric);
   Mark(1);
   const int DoDel2Tr = (NT > 0 && P.TracerHyperDiffTendencyEnable) ? 1 : 0;
   bool Cell1Done = false;
   // the merged kernel can take the side-0 sums of the cells with one edge fewer than the sweep's valence along (INLO)
   const bool InlineOther = MergeL1 && ND == TME && NOther > 0;
   if (MergeL1) {
      auto LaunchL1x = [&](auto Epi, auto Inl) {
         constexpr bool EP = decltype(Epi)::value, IL = decltype(Inl)::value && ND == TME;
         FusedCellL1PVBody<TME, Fast, EP, ND, IL, FLS> B{M,  K,  NT,    P,      DoDel2Tr,        H,
                                            U,  Tr, A.KineticEnergyCell, A.VelocityDivCell, HTend, A.Del2TracersCell,
                                            A.RelVortVertex, A.InvThickVertex, EdgeScratch, EH};
         if constexpr (CanWide) {
            if (NWide > 0) { // the wide cells' level-1 work rides along: same body, TW slots, wide tables, cell list
               FusedCellL1PVBody<TW, Fast, EP, TW, false, 1> Bw{*Wide, K,  NT,    P,      DoDel2Tr,        H,
                                                       U,     Tr, A.KineticEnergyCell, A.VelocityDivCell, HTend, A.Del2TracersCell,
                                                       A.RelVortVertex, A.InvThickVertex, EdgeScratch, EH};
               Bw.List = Wide->WideCells;
               launchTileV(K, S, B, NSweepL1, Bw, NWide);
               return;
            }
         }
         B.SkipBad = M.NBadCells > 0;
         if (M.NBadCells > 0) {
            // the cells outside the ring tables: the generic level-1 cell body over their list, in the sweep's launch
            // (their edges are on the irregular-edge list); the vertices no good cell stores through the vertex kernel
            FusedCell1Body<TME, Fast, EP> Bb{M, K, NT, P, DoDel2Tr, H, U, Tr, A.KineticEnergyCell, A.VelocityDivCell,
                                             HTend, A.Del2TracersCell, EH};
            Bb.List = M.BadCells;
            launchTileV(K, S, B, NSweepL1, Bb, M.NBadCells);
            launchVertexAuxState1List(M, K, A, H, U, S, M.OrphanVertices, M.NOrphanVertices);
            return;
         }
         launchTile(B, NSweepL1, K, S);
      };
      auto LaunchL1 = [&](auto Epi) {
         if (InlineOther)
            LaunchL1x(Epi, std::true_type{});
         else
            LaunchL1x(Epi, std::false_type{});
      };
      if constexpr (Fast) {
         if (Stage)
            LaunchL1(std::true_type{});
         else
            LaunchL1(std::false_type{});
      } else {
         LaunchL1(std::false_type{});
      }
      Cell1Done = true;
   }
   if constexpr (Fast) {
      if (Stage && !Cell1Done) {
         FusedCell1Body<TME, true, true> B{M, K, NT, P, DoDel2Tr, H, U, Tr, A.KineticEnergyCell, A.VelocityDivCell,
                                           HTend, A.Del2TracersCell, EH};
         launchTile(B, M.NCellsAll, K, S);
         Cell1Done = true;
      }
   }
   if (!Cell1Done) {
      FusedCell1Body<TME, Fast> B{M, K, NT, P, DoDel2Tr, H, U, Tr, A.KineticEnergyCell, A.VelocityDivCell, HTend,
                                  A.Del2TracersCell};
      launchTile(B, M.NCellsAll, K, S);
   }
   if constexpr (CanWide) {
      if (NWide > 0 && !MergeL1) { // the wide cells' level-1 work (merged kernel: launched together with the sweep above)
         auto WideL1 = [&](auto Epi) {
            constexpr bool EP = decltype(Epi)::value;
            FusedCell1Body<TW, Fast, EP> B{*Wide, K, NT, P, DoDel2Tr, H, U, Tr, A.KineticEnergyCell, A.VelocityDivCell,
                                           HTend, A.Del2TracersCell, EH};
            B.List = Wide->WideCells;
            launchTile(B, NWide, K, S);
         };
         if constexpr (Fast) {
            if (Stage)
               WideL1(std::true_type{});
            else
               WideL1(std::false_type{});
         } else {
            WideL1(std::false_type{});
         }
      }
   }
   if (P.WindForcingTendencyEnable)
      launchEdgeAuxState1(Wide ? *Wide : M, A, P.WindInterpIsotropic, S);
   Pacer::stop("Tend:fused:L1", 2);
   // L2 (only the del4 term consumes it): replaces AuxState:edgeAuxState3 (Del2Edge), cellAuxState2, vertexAuxState2
   Pacer::start("Tend:fused:L2[AuxState:vertexAuxState2,cellAuxState2]", 2);
   Mark(2);
   // independent sweeps share a launch (KernelCommon.h: tileKernel2); option Pair = 0 launches them one by one
   const int PairEnv = Tn.Pair;
   const bool PairL2        = PairEnv && P.VelHyperDiffTendencyEnable && M.Del2RingOK && M.Del2VertOK;
   bool WideL2Done = false, BadL2Done = false;
   FusedKernelNames[2] = FusedKernelNames[3] = "";
   if (PairL2) {
      FusedKernelNames[2] = "Del2CellRingBody+Del2VertexSelBody";
      Del2CellRingBody<TME, FLS> BC{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2DivCell};
      Del2VertexSelBody BV{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2RelVortVertex};
      bool Launched = false;
      if constexpr (CanWide) {
         if (NWide > 0) {
            Del2CellRingBody<TW, 1> BW{*Wide, K, A.VelocityDivCell, A.RelVortVertex, A.Del2DivCell};
            BW.List = Wide->WideCells;
            launchTileV(K, S, BC, M.NCellsAll, BV, M.NVerticesAll, BW, NWide);
            Launched = WideL2Done = true;
         }
      }
      if (!Launched && M.NBadCells > 0) { // (the cells outside the ring tables ride along: generic body over their list)
         FusedDel2CellBody Bb{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2DivCell, M.BadCells};
         launchTileV(K, S, BC, M.NCellsAll, BV, M.NVerticesAll, Bb, M.NBadCells);
         Launched = BadL2Done = true;
      }
      if (!Launched)
         launchTile2(BC, M.NCellsAll, BV, M.NVerticesAll, K, S);
   } else if (P.VelHyperDiffTendencyEnable) {
      FusedKernelNames[2] = M.Del2RingOK ? "Del2CellRingBody" : "FusedDel2CellBody";
      if (M.Del2RingOK) {
         Del2CellRingBody<TME, FLS> BC{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2DivCell};
            launchTile(BC, M.NCellsAll, K, S);
      } else {
         FusedDel2CellBody BC{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2DivCell};
         launchTile(BC, M.NCellsAll, K, S);
      }
   }
   Mark(3);
   if (P.VelHyperDiffTendencyEnable && !PairL2) {
      FusedKernelNames[3] = M.Del2VertOK ? "Del2VertexSelBody" : "FusedDel2VertexBody";
      if (M.Del2VertOK) {
         Del2VertexSelBody BV{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2RelVortVertex};
         launchTile(BV, M.NVerticesAll, K, S);
      } else {
         FusedDel2VertexBody BV{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2RelVortVertex};
         launchTile(BV, M.NVerticesAll, K, S);
      }
   }
   if constexpr (CanWide) {
      if (NWide > 0 && P.VelHyperDiffTendencyEnable && !WideL2Done) { // (a narrow view implies the ring form)
         Del2CellRingBody<TW, 1> BC{*Wide, K, A.VelocityDivCell, A.RelVortVertex, A.Del2DivCell};
         BC.List = Wide->WideCells;
         launchTile(BC, NWide, K, S);
      }
   }
   if (P.VelHyperDiffTendencyEnable && M.Del2RingOK && M.NBadCells > 0 && !BadL2Done) { // the cells outside the ring tables
      FusedDel2CellBody Bb{M, K, A.VelocityDivCell, A.RelVortVertex, A.Del2DivCell, M.BadCells};
      launchTile(Bb, M.NBadCells, K, S);
   }
   Pacer::stop("Tend:fused:L2", 2);
   // L3: replaces Tend:potientialVortHAdv, KEGrad, SSHGrad, velocityDiffusion, velocityHyperDiff, windForcing, bottomDrag,
   // AuxState:edgeAuxState4 (HTracersEdge) and Tend:tracerHorzAdv, tracerDiffusion, tracerHyperDiff
   Pacer::start("Tend:fused:L3[Tend:potientialVortHAdv,KEGrad,SSHGrad,velocityDiffusion,velocityHyperDiff,tracerHorzAdv,"
                "tracerDiffusion,tracerHyperDiff]", 2);
   Mark(4);
   bool Marked5        = false;
   std::function<void()> LaunchFinalInterior; // set when the side-1 sweep is split for an overlapped exchange
   FusedKernelNames[4] = "FusedEdgeChainBody", FusedKernelNames[5] = "";
   // the side-1 PV + velocity kernel and the tracer kernel are independent: their main sweeps share a launch
   const bool PairL3 = PairEnv && Fast && M.CellPVOK && EdgeScratch && P.PVTendencyEnable && M.CellPVFinalOK && NT > 0 &&
                       NMain > 0;
   // the plain RHS does both in ONE thread per (cell, levels): h and u gathered once (CellPVFinalTracerBody)
   const bool FuseL3 = PairL3 && !Stage;
   // narrow tables, plain RHS: the wide cells' level-3 work (one thread does velocity + tracers, as the sweep's) and the
   // final pass of the other valence's list join the sweep's launch instead of being launches of their own
   // (measured on a QU240-sized sphere, 12 pentagons: their final-pass list inside the sweep's launch: RHS 109 -> 102 us;
   // the same for the stage pair, as a third body, and the side-0 list folded into the level-2 launch: both slower)
   const bool FoldL3 = FuseL3 && ((CanWide && NWide > 0) || NOther > 0);
   // plain RHS, one table width: the irregular-edge list (coast lines; the masked rim of a partition's halo) joins the
   // sweep's launch too (an eighth of the QU30-sized mesh with its halo: one launch of ~10 us less per RHS)
   const bool FoldChain = FuseL3 && !Wide && M.NIrregularEdges > 0;
   if (M.CellPVOK && EdgeScratch) {
      const bool PVOn = P.PVTendencyEnable != 0;
      bool Finished   = false;
      FusedKernelNames[4] = "";
      if (PVOn) {
         // the rarer valences (MaxEdges-1, MaxEdges-2: e.g. the pentagons of a mesh stored with
         // maxEdges = 6 or 7) run the same ring code, instantiated for their size, over cell lists
         constexpr int NM1 = NA, NM2 = TME >= 6 ? TME - 2 : TME - 1; // (NM1: "the other big valence")
         CellPVBody<TME, Fast, 0, ND> B0{M, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch};
         if (NMain > 0 && !MergeL1) // (merged: done by the L1 kernel; only the rarer valences remain)
            launchTile(B0, M.NCellsAll, K, S);
         FusedKernelNames[4] =
             (!MergeL1 || (NOther > 0 && !InlineOther) || (TME >= 6 && M.NRingCellsM2 > 0)) ? "CellPVBody<side 0>" : "";
         if (NOther > 0 && !InlineOther) {
            CellPVBody<TME, Fast, 0, NM1> Bm{M, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                             OtherCells};
            launchTile(Bm, NOther, K, S);
         }
         if (TME >= 6 && M.NRingCellsM2 > 0) {
            CellPVBody<TME, Fast, 0, NM2> Bm{M, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                             M.RingCellsM2};
            launchTile(Bm, M.NRingCellsM2, K, S);
         }
         if constexpr (CanWide) {
            if (NWide > 0 && !MergeL1) {
               CellPVBody<TW, Fast, 0, TW> Bw{*Wide, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                              Wide->WideCells};
               launchTile(Bw, NWide, K, S);
            }
         }
         Mark(5);
         Marked5 = true;
         if (Fast && M.CellPVFinalOK) {
            // Overlap == true: the full sweep is split into the band list now and the interior list after
            // the exchange has been started (Stage->AfterBand), see Kernels.h: StageUpdate
            const bool Overlap = Stage && Stage->AfterBand && M.NBandCells > 0;
            auto LaunchFinal = [&](auto Epi) {
               constexpr bool EP = decltype(Epi)::value;
               CellPVFinalBody<TME, ND, EP> B1{M,
                                                K,
                                                P,
                                                H,
                                                U,
                                                A.RelVortVertex,
                                                A.InvThickVertex,
                                                EdgeScratch,
                                                A.RelVortVertex,
                                                A.KineticEnergyCell,
                                                A.VelocityDivCell,
                                                A.Del2DivCell,
                                                A.Del2RelVortVertex,
                                                UTend,
                                                nullptr,
                                                EU};
               if (NMain > 0 && !PairL3) { // (paired: launched together with the tracer kernel below)
                  if (Overlap) {
                     B1.List = BandList;
                     launchTile(B1, NBandList, K, BandS());
                  } else {
                     launchTile(B1, EP ? NSweepVel : M.NCellsAll, K, S);
                  }
               }
               if (NOther > 0 && !FoldL3) {
                  CellPVFinalBody<TME, NM1, EP> Bm{B1.M,       B1.K,   B1.P,    B1.H,       B1.U,
                                                   B1.RelVortV, B1.InvThickV, B1.Partial, B1.RelVort, B1.KE,
                                                   B1.Div,     B1.Del2Div, B1.Del2RelVort, B1.Tend, OtherCells, EU};
                  launchTile(Bm, NOther, K, S);
               }
               if (TME >= 6 && M.NRingCellsM2 > 0) {
                  CellPVFinalBody<TME, NM2, EP> Bm{B1.M,       B1.K,   B1.P,    B1.H,       B1.U,
                                                   B1.RelVortV, B1.InvThickV, B1.Partial, B1.RelVort, B1.KE,
                                                   B1.Div,     B1.Del2Div, B1.Del2RelVort, B1.Tend, M.RingCellsM2, EU};
                  launchTile(Bm, M.NRingCellsM2, K, S);
               }
               if constexpr (CanWide) {
                  if (NWide > 0 && !FoldL3) {
                     CellPVFinalBody<TW, TW, EP> Bw{*Wide,      B1.K,   B1.P,    B1.H,       B1.U,
                                                    B1.RelVortV, B1.InvThickV, B1.Partial, B1.RelVort, B1.KE,
                                                    B1.Div,     B1.Del2Div, B1.Del2RelVort, B1.Tend, Wide->WideCells, EU};
                     launchTile(Bw, NWide, K, S);
                  }
               }
            };
            // the interior part of the split sweep, launched at the end of the L3 phase
            LaunchFinalInterior = [&, Overlap]() {
               (void)Overlap;
               if constexpr (Fast) {
                  if (Overlap && NMain > 0 && M.NInteriorCells > 0 && !PairL3) {
                     CellPVFinalBody<TME, ND, true> B1{M,
                                                        K,
                                                        P,
                                                        H,
                                                        U,
                                                        A.RelVortVertex,
                                                        A.InvThickVertex,
                                                        EdgeScratch,
                                                        A.RelVortVertex,
                                                        A.KineticEnergyCell,
                                                        A.VelocityDivCell,
                                                        A.Del2DivCell,
                                                        A.Del2RelVortVertex,
                                                        UTend,
                                                        M.InteriorCells,
                                                        EU};
                     launchTile(B1, M.NInteriorCells, K, S);
                  }
               }
            };
            if (Stage)
               LaunchFinal(std::true_type{});
            else
               LaunchFinal(std::false_type{});
            Finished            = true;
            // (paired: the main sweep runs in slot 6 together with the tracer kernel; only the list launches of the
            // rarer valences remain here)
            FusedKernelNames[5] = !PairL3 ? "CellPVFinalBody"
                                          : ((NOther > 0 || (TME >= 6 && M.NRingCellsM2 > 0)) ? "CellPVFinalBody (rarer valences)" : "");
         } else {
            CellPVBody<TME, Fast, 1, ND> B1{M, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch};
            if (NMain > 0)
               launchTile(B1, M.NCellsAll, K, S);
            if (NOther > 0) {
               CellPVBody<TME, Fast, 1, NM1> Bm{M, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                                OtherCells};
               launchTile(Bm, NOther, K, S);
            }
            if (TME >= 6 && M.NRingCellsM2 > 0) {
               CellPVBody<TME, Fast, 1, NM2> Bm{M, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                                M.RingCellsM2};
               launchTile(Bm, M.NRingCellsM2, K, S);
            }
            if constexpr (CanWide) {
               if (NWide > 0) {
                  CellPVBody<TW, Fast, 1, TW> Bw{*Wide, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                                 Wide->WideCells};
                  launchTile(Bw, NWide, K, S);
               }
            }
         }
      }
      if (!Finished) {
         FusedKernelNames[5] = "CellPVBody<side 1>+EdgeFinalBody";
         EdgeFinalBody<Fast> BF{M,
                                K,
                                P,
                                H,
                                U,
                                EdgeScratch,
                                A.RelVortVertex,
                                A.KineticEnergyCell,
                                A.VelocityDivCell,
                                A.Del2DivCell,
                                A.Del2RelVortVertex,
                                A.NormalStressEdge,
                                UTend};
         launchTile(BF, M.NEdgesAll, K, S);
      }
      // (a stage whose halo outputs the exchange replaces: the owned irregular edges -- a coast -- only, not the masked
      // edges of the halo rim)
      // (... and a stage whose velocity sweep stops after halo layer 3 finishes the edges of the cells through layer 2)
      const int NIrr = SendOnly ? M.NIrregularOwned : (NSweepVel < M.NCellsAll ? M.NIrregularInner : M.NIrregularEdges);
      if (NIrr > 0 && !FoldChain) {
         auto LaunchList = [&](auto Epi) {
            constexpr bool EP = decltype(Epi)::value;
            if constexpr (CanWide) {
               if (Wide) { // (the chain tables are per edge and MaxEdges of the WIDE view wide)
                  FusedEdgeChainBody<TW, Fast, EP, true> B{*Wide, K, P, H, U, A.RelVortVertex, A.InvThickVertex, nullptr,
                                                           A.KineticEnergyCell, A.VelocityDivCell, A.Del2DivCell,
                                                           A.Del2RelVortVertex, A.NormalStressEdge, UTend,
                                                           M.IrregularEdges, EU};
                  launchTile(B, NIrr, K, S);
                  return;
               }
            }
            FusedEdgeChainBody<TME, Fast, EP, true> B{M,
                                                      K,
                                                      P,
                                                      H,
                                                      U,
                                                      A.RelVortVertex,
                                                      A.InvThickVertex,
                                                      nullptr,
                                                      A.KineticEnergyCell,
                                                A.VelocityDivCell,
                                                A.Del2DivCell,
                                                A.Del2RelVortVertex,
                                                A.NormalStressEdge,
                                                UTend,
                                                M.IrregularEdges,
                                                EU};
            launchTile(B, NIrr, K, S);
         };
         if (Stage)
            LaunchList(std::true_type{});
         else
            LaunchList(std::false_type{});
      }
   } else if (M.PVChainOK) {
      FusedEdgeChainBody<TME, Fast> B{M,
                                      K,
                                      P,
                                      H,
                                      U,
                                      A.RelVortVertex,
                                      A.NormRelVortVertex,
                                      A.NormPlanetVortVertex,
                                      A.KineticEnergyCell,
                                      A.VelocityDivCell,
                                      A.Del2DivCell,
                                      A.Del2RelVortVertex,
                                      A.NormalStressEdge,
                                      UTend,
                                      nullptr};
      launchTile(B, M.NEdgesAll, K, S);
   } else {
      FusedKernelNames[4] = "FusedEdgeBody";
      FusedEdgeBody B{M,       K,           P,           H,           U,
                      A.RelVortVertex, A.NormRelVortVertex, A.NormPlanetVortVertex, A.KineticEnergyCell, A.VelocityDivCell,
                      A.Del2DivCell,   A.Del2RelVortVertex, A.NormalStressEdge,     UTend};
      launchTile(B, M.NEdgesAll, K, S);
   }
   if (!Marked5)
      Mark(5);
   Mark(6);
   FusedKernelNames[6]        = FuseL3   ? "CellPVFinalTracerBody"
                                : PairL3 ? "CellPVFinalBody+FusedCell3Body"
                                         : (NT > 0 ? "FusedCell3Body" : "");
   if constexpr (CanWide) {
      if (NWide > 0 && NT > 0 && !FoldL3) {
         auto WideTr = [&](auto Epi) {
            constexpr bool EP = decltype(Epi)::value;
            FusedCell3Body<TW, Fast, EP, 1> B{*Wide, K, NT, P, H, U, Tr, A.Del2TracersCell, TrTend, ET};
            B.List = Wide->WideCells;
            launchTile(B, NWide, K, S);
         };
         if constexpr (Fast) {
            if (Stage)
               WideTr(std::true_type{});
            else
               WideTr(std::false_type{});
         } else {
            WideTr(std::false_type{});
         }
      }
   }
   bool AfterBandCalled = false;
   if (PairL3) {
      if constexpr (Fast) {
         auto Go = [&](auto Epi) {
            constexpr bool EP = decltype(Epi)::value;
            CellPVFinalBody<TME, ND, EP> B1{M,
                                             K,
                                             P,
                                             H,
                                             U,
                                             A.RelVortVertex,
                                             A.InvThickVertex,
                                             EdgeScratch,
                                             A.RelVortVertex,
                                             A.KineticEnergyCell,
                                             A.VelocityDivCell,
                                             A.Del2DivCell,
                                             A.Del2RelVortVertex,
                                             UTend,
                                             nullptr,
                                             EU};
            FusedCell3Body<TME, true, EP, FLL> B3{M, K, NT, P, H, U, Tr, A.Del2TracersCell, TrTend, ET};
            if constexpr (!EP) {
               { // plain RHS (FuseL3): one thread per (cell, levels) does both, h and u gathered once
                  CellPVFinalTracerBody<TME, ND, FLS> BF{M, K, NT, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                                A.RelVortVertex, A.KineticEnergyCell, A.VelocityDivCell, A.Del2DivCell,
                                                A.Del2RelVortVertex, UTend, Tr, A.Del2TracersCell, TrTend};
                  // the sweep's body Bs -- plain or with the tracer loop through LDS tile patches -- alone or with the lists
                  // that ride along in its launch
                  auto LaunchSweep = [&](const auto &Bs) {
                  if constexpr (CanWide) {
                        if (FoldL3 && NWide > 0) {
                           CellPVFinalTracerBody<TW, TW, 1> BW{*Wide, K, NT, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                                            A.RelVortVertex, A.KineticEnergyCell, A.VelocityDivCell, A.Del2DivCell,
                                                            A.Del2RelVortVertex, UTend, Tr, A.Del2TracersCell, TrTend};
                           BW.List = Wide->WideCells;
                           constexpr int NM1f = ND == TME ? TME - 1 : TME;
                           CellPVFinalBody<TME, NM1f, false> Bm{M, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                                                 A.RelVortVertex, A.KineticEnergyCell, A.VelocityDivCell,
                                                                 A.Del2DivCell, A.Del2RelVortVertex, UTend, OtherCells, EU};
                           launchTileV(K, S, Bs, M.NCellsAll, BW, NWide, Bm, NOther);
                           return;
                        }
                     }
                     if (FoldL3 || FoldChain) { // (no wide cells: the sweep, the other valence's final pass, the irregular edges)
                        constexpr int NM1f = ND == TME ? TME - 1 : TME;
                        CellPVFinalBody<TME, NM1f, false> Bm{M, K, P, H, U, A.RelVortVertex, A.InvThickVertex, EdgeScratch,
                                                              A.RelVortVertex, A.KineticEnergyCell, A.VelocityDivCell,
                                                              A.Del2DivCell, A.Del2RelVortVertex, UTend, OtherCells, EU};
                        FusedEdgeChainBody<TME, Fast, false, true> Bc{M, K, P, H, U, A.RelVortVertex, A.InvThickVertex, nullptr,
                                                                      A.KineticEnergyCell, A.VelocityDivCell, A.Del2DivCell,
                                                                      A.Del2RelVortVertex, A.NormalStressEdge, UTend,
                                                                      M.IrregularEdges, EU};
                        const int NO = FoldL3 ? NOther : 0, NC = FoldChain ? M.NIrregularEdges : 0;
                        if (NO > 0 && NC > 0)
                           launchTileV(K, S, Bs, M.NCellsAll, Bm, NO, Bc, NC);
                        else if (NC > 0)
                           launchTileV(K, S, Bs, M.NCellsAll, Bc, NC);
                        else
                           launchTileV(K, S, Bs, M.NCellsAll, Bm, NO);
                        return;
                     }
                     launchTile(Bs, M.NCellsAll, K, S);
                     return;
                  };
                  if constexpr (TME <= 7) {
                     // option TracerPatch: the tracer loop's neighbour values through LDS tile patches (16-byte accesses,
                     // line-wide thread geometry and a tile size the mesh has patch tables for); the lists keep their bodies
                     const int NList = (CanWide && FoldL3 ? NWide : 0) + (FoldL3 ? NOther : 0) + (FoldChain ? M.NIrregularEdges : 0);
                     const Geom Gp   = makeGeom(M.NCellsAll + NList, K, 2, levelPitch(K), NT <= 8 ? 16 : 0);
                     const int Slot  = MeshView::patchSlot(Gp.Tile);
                     // (from 4 tracers on: with 2 the transfers' set-up and the barriers cost more than they save --
                     // EC30to60-sized, 2 tracers: level 3 +1.7 %, QU240-sized +7 %; an eighth of QU30, 6 tracers: -2.3 %)
                     if (Tn.TracerPatch && NT >= 4 && Gp.W == 2 && Gp.Block.x == 8 && Slot >= 0 && (int)Gp.Block.y == Gp.Tile) {
                        CellPVFinalTracerPatchBody<TME, ND, FLS> BP{{BF}, M.PatchRows[Slot], M.PatchIdx[Slot], M.PatchOK[Slot],
                                                                   M.PatchNP[Slot], Gp.Tile};
                        LaunchSweep(BP);
                        return;
                     }
                  }
                  LaunchSweep(BF);
                  return;
               }
            } else if (Stage->AfterBand && M.NBandCells > 0) { // RK4 stages: the two bodies as a paired launch
               B1.List = B3.List = BandList;
               launchTile2(B1, NBandList, B3, NBandList, K, BandS());
               Stage->AfterBand(Stage->AfterBandCtx); // u, h and the tracers of every sent element are final
               AfterBandCalled = true;
               B1.List = B3.List = M.InteriorCells;
               launchTile2(B1, M.NInteriorCells, B3, M.NInteriorCells, K, S);
            } else {
               launchTile2(B1, NSweepVel, B3, NSweepTr, K, S);
            }
         };
         if (Stage)
            Go(std::true_type{});
         else
            Go(std::false_type{});
      }
   } else if (NT > 0) {
      bool Done = false;
      if constexpr (Fast) {
         if (Stage) {
            FusedCell3Body<TME, true, true, FLL> B{M, K, NT, P, H, U, Tr, A.Del2TracersCell, TrTend, ET};
            if (Stage->AfterBand && M.NBandCells > 0) {
               B.List = BandList;
               launchTile(B, NBandList, K, BandS());
               Stage->AfterBand(Stage->AfterBandCtx); // u, h and the tracers of every sent element are final
               AfterBandCalled = true;
               if (LaunchFinalInterior)
                  LaunchFinalInterior();
               B.List = M.InteriorCells;
               launchTile(B, M.NInteriorCells, K, S);
            } else {
               launchTile(B, NSweepTr, K, S);
            }
            Done = true;
         }
      }
      if (!Done) {
         FusedCell3Body<TME, Fast, false, FLL> B{M, K, NT, P, H, U, Tr, A.Del2TracersCell, TrTend};
         launchTile(B, M.NCellsAll, K, S);
      }
   }
   if (Stage && Stage->AfterBand && !AfterBandCalled) { // no tracers (or no band): everything is final here
      Stage->AfterBand(Stage->AfterBandCtx);
      if (LaunchFinalInterior)
         LaunchFinalInterior();
   }
   Mark(7);
   Pacer::stop("Tend:fused:L3", 2);
}


/// the signature of launchFusedT, for the explicit instantiations (FusedInst*.hip) and their declarations (FusedKernels.hip)
#define OMEGA_FUSED_ARGS                                                                                           \
   (const MeshView &, int, int, const TendParams &, const AuxPtrs &, Real *, Real *, Real *, const Real *, const Real *,      \
    const Real *, hipStream_t, hipEvent_t *, Real *, const StageUpdate *, const MeshView *)
/// every instantiation the dispatcher calls, in the groups the translation units compile: X(TME, Fast, ND, HW)
#ifdef OMEGA_ONLY_ME6 // (measurement builds of a kernel experiment: hexagon meshes only, a quarter of the compile time)
#define OMEGA_FUSED_INSTANCES_5(X)
#define OMEGA_FUSED_INSTANCES_5N(X)
#define OMEGA_FUSED_INSTANCES_6N(X)
#define OMEGA_FUSED_INSTANCES_7(X)
#define OMEGA_FUSED_INSTANCES_7N(X)
#define OMEGA_FUSED_INSTANCES_8(X)
#else
#define OMEGA_FUSED_INSTANCES_5(X) X(5, true, 5, false) X(5, false, 5, false)
#define OMEGA_FUSED_INSTANCES_5N(X) X(5, true, 5, true) X(5, false, 5, true)
#define OMEGA_FUSED_INSTANCES_6N(X) X(6, true, 6, true) X(6, false, 6, true)
#define OMEGA_FUSED_INSTANCES_7(X) X(7, true, 6, false) X(7, true, 7, false) X(7, false, 7, false)
#define OMEGA_FUSED_INSTANCES_7N(X) X(7, true, 7, true) X(7, false, 7, true)
#define OMEGA_FUSED_INSTANCES_8(X) X(8, true, 7, false) X(8, true, 8, false) X(8, false, 8, false)
#endif
#define OMEGA_FUSED_INSTANCES_6A(X) X(6, true, 5, false) X(6, true, 6, false)
#define OMEGA_FUSED_INSTANCES_6B(X) X(6, false, 6, false)
#define OMEGA_FUSED_DEFINE(TME_, FAST_, ND_, HW_) template void launchFusedT<TME_, FAST_, ND_, HW_> OMEGA_FUSED_ARGS;
#define OMEGA_FUSED_DECLARE(TME_, FAST_, ND_, HW_) extern template void launchFusedT<TME_, FAST_, ND_, HW_> OMEGA_FUSED_ARGS;

} // namespace OMEGA
#endif
