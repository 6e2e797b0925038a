// HorzMesh.cpp -- see HorzMesh.h.
#include "HorzMesh.h"
#include "Tuning.h"

#include <cstdlib>

#include <algorithm>
#include <type_traits>
#include <cmath>

namespace OMEGA {

namespace {
// gather a global per-element array (Width values per element) into local order;
// the sentinel row stays zero (the reference reads only NXxAll elements into
// NXxSize-long zero-initialised arrays, HorzMesh.cpp:360-420)
HostArrayReal gatherLocal(const R8 *Global, const HostArrayI4 &IDH, I4 NAll, int Width = 1) {
   HostArrayReal Out(NAll + 1, Width, 1, 0.0);
   if (!Global)
      return Out;
   for (I4 L = 0; L < NAll; ++L) {
      const size_t Gid = (size_t)(IDH(L) - 1);
      for (int J = 0; J < Width; ++J)
         Out.V[(size_t)L * Width + J] = Global[Gid * Width + J];
   }
   return Out;
}
} // namespace

HorzMesh::HorzMesh(const std::string &Name, const Decomp *D, I4 InNVertLayers, bool HostOnly_) {
   HostOnly    = HostOnly_;
   MeshName    = Name;
   NVertLayers = InNVertLayers;
   OMEGA_REQUIRE(NVertLayers >= 1, "HorzMesh: NVertLayers must be >= 1");

   NCellsHaloH = D->NCellsHaloH;
   NCellsOwned = D->NCellsOwned;
   NCellsAll   = D->NCellsAll;
   NCellsSize  = D->NCellsSize;

   NEdgesHaloH    = D->NEdgesHaloH;
   NEdgesOwned    = D->NEdgesOwned;
   NEdgesAll      = D->NEdgesAll;
   NEdgesSize     = D->NEdgesSize;
   MaxCellsOnEdge = D->MaxCellsOnEdge;
   MaxEdges       = D->MaxEdges;
   MaxEdges2      = 2 * MaxEdges;

   NVerticesHaloH = D->NVerticesHaloH;
   NVerticesOwned = D->NVerticesOwned;
   NVerticesAll   = D->NVerticesAll;
   NVerticesSize  = D->NVerticesSize;
   VertexDegree   = D->VertexDegree;

   CellsOnCellH    = D->CellsOnCellH;
   EdgesOnCellH    = D->EdgesOnCellH;
   NEdgesOnCellH   = D->NEdgesOnCellH;
   VerticesOnCellH = D->VerticesOnCellH;
   CellsOnEdgeH    = D->CellsOnEdgeH;
   EdgesOnEdgeH    = D->EdgesOnEdgeH;
   NEdgesOnEdgeH   = D->NEdgesOnEdgeH;
   VerticesOnEdgeH = D->VerticesOnEdgeH;
   CellsOnVertexH  = D->CellsOnVertexH;
   EdgesOnVertexH  = D->EdgesOnVertexH;

   const GlobalMeshDesc &G = D->globalMesh();
   // readCoordinates (HorzMesh.cpp:424-449)
   XCellH     = gatherLocal(G.XCell, D->CellIDH, NCellsAll);
   YCellH     = gatherLocal(G.YCell, D->CellIDH, NCellsAll);
   ZCellH     = gatherLocal(G.ZCell, D->CellIDH, NCellsAll);
   LonCellH   = gatherLocal(G.LonCell, D->CellIDH, NCellsAll);
   LatCellH   = gatherLocal(G.LatCell, D->CellIDH, NCellsAll);
   XEdgeH     = gatherLocal(G.XEdge, D->EdgeIDH, NEdgesAll);
   YEdgeH     = gatherLocal(G.YEdge, D->EdgeIDH, NEdgesAll);
   ZEdgeH     = gatherLocal(G.ZEdge, D->EdgeIDH, NEdgesAll);
   LonEdgeH   = gatherLocal(G.LonEdge, D->EdgeIDH, NEdgesAll);
   LatEdgeH   = gatherLocal(G.LatEdge, D->EdgeIDH, NEdgesAll);
   XVertexH   = gatherLocal(G.XVertex, D->VertexIDH, NVerticesAll);
   YVertexH   = gatherLocal(G.YVertex, D->VertexIDH, NVerticesAll);
   ZVertexH   = gatherLocal(G.ZVertex, D->VertexIDH, NVerticesAll);
   LonVertexH = gatherLocal(G.LonVertex, D->VertexIDH, NVerticesAll);
   LatVertexH = gatherLocal(G.LatVertex, D->VertexIDH, NVerticesAll);
   // readBottomDepth / readMeasurements / readWeights / readCoriolis (:453-523)
   OMEGA_REQUIRE(G.AreaCell && G.AreaTriangle && G.KiteAreasOnVertex && G.DcEdge && G.DvEdge && G.WeightsOnEdge,
                 "HorzMesh: global mesh geometry incomplete");
   BottomDepthH       = gatherLocal(G.BottomDepth, D->CellIDH, NCellsAll);
   AreaCellH          = gatherLocal(G.AreaCell, D->CellIDH, NCellsAll);
   AreaTriangleH      = gatherLocal(G.AreaTriangle, D->VertexIDH, NVerticesAll);
   DvEdgeH            = gatherLocal(G.DvEdge, D->EdgeIDH, NEdgesAll);
   DcEdgeH            = gatherLocal(G.DcEdge, D->EdgeIDH, NEdgesAll);
   AngleEdgeH         = gatherLocal(G.AngleEdge, D->EdgeIDH, NEdgesAll);
   KiteAreasOnVertexH = gatherLocal(G.KiteAreasOnVertex, D->VertexIDH, NVerticesAll, VertexDegree);
   WeightsOnEdgeH     = gatherLocal(G.WeightsOnEdge, D->EdgeIDH, NEdgesAll, MaxEdges2);
   compactMaxEdges();
   FCellH             = gatherLocal(G.FCell, D->CellIDH, NCellsAll);
   FVertexH           = gatherLocal(G.FVertex, D->VertexIDH, NVerticesAll);
   FEdgeH             = gatherLocal(G.FEdge, D->EdgeIDH, NEdgesAll);

   computeEdgeSign();
   setMasks();
   setMeshScaling();
   if (!HostOnly) {
      copyToDevice();
      buildCoefficientTables();
   }
}

// Mesh files often carry a maxEdges dimension larger than any cell uses (7, 8, 10 for a mesh of pentagons and
// hexagons).  The kernels are instantiated for the table width, so the mesh keeps its cell-slot tables at the
// largest valence actually present on this rank (never below 5): MaxEdges / MaxEdges2 of a HorzMesh are THAT width,
// Decomp keeps the file's.  Padding slots never take part in any result.
void HorzMesh::compactMaxEdges() {
   MaxEdgesFile = MaxEdges;
   // option KeepMaxEdges (test hook, Tuning.h): keep the file's width, so that the wide kernel instantiations and the
   // edge-centric list for valences below MaxEdges-2 can be exercised with meshes that have no such cells
   if (tuning().KeepMaxEdges != 0)
      return;
   int Eff      = 5;
   for (int C = 0; C < NCellsAll; ++C)
      Eff = std::max(Eff, (int)NEdgesOnCellH(C));
   int Eff2 = 0; // EdgesOnEdge rows hold their entries contiguously from slot 0
   for (int E = 0; E < NEdgesAll; ++E)
      Eff2 = std::max(Eff2, (int)NEdgesOnEdgeH(E));
   Eff = std::max(Eff, (Eff2 + 1) / 2);
   if (Eff >= MaxEdges)
      return;
   auto Narrow = [](const auto &A, int NewW) {
      std::decay_t<decltype(A)> R(A.Ext[0], NewW);
      for (int I = 0; I < A.Ext[0]; ++I)
         for (int J = 0; J < NewW; ++J)
            R(I, J) = A(I, J);
      return R;
   };
   CellsOnCellH    = Narrow(CellsOnCellH, Eff);
   EdgesOnCellH    = Narrow(EdgesOnCellH, Eff);
   VerticesOnCellH = Narrow(VerticesOnCellH, Eff);
   EdgesOnEdgeH    = Narrow(EdgesOnEdgeH, 2 * Eff);
   WeightsOnEdgeH  = Narrow(WeightsOnEdgeH, 2 * Eff);
   MaxEdges        = Eff;
   MaxEdges2       = 2 * Eff;
}

// HorzMesh::computeEdgeSign (reference HorzMesh.cpp:527-575)
void HorzMesh::computeEdgeSign() {
   EdgeSignOnCellH = HostArrayReal(NCellsSize, MaxEdges, 1, 0.0);
   for (int Cell = 0; Cell < NCellsAll; ++Cell)
      for (int I = 0; I < NEdgesOnCellH(Cell); ++I) {
         int Edge = EdgesOnCellH(Cell, I);
         // vector points from cell 0 to cell 1
         EdgeSignOnCellH(Cell, I) = (Cell == CellsOnEdgeH(Edge, 0)) ? -1.0 : 1.0;
      }
   EdgeSignOnVertexH = HostArrayReal(NVerticesSize, VertexDegree, 1, 0.0);
   for (int Vertex = 0; Vertex < NVerticesAll; ++Vertex)
      for (int I = 0; I < VertexDegree; ++I) {
         int Edge = EdgesOnVertexH(Vertex, I);
         // vector points from vertex 0 to vertex 1
         EdgeSignOnVertexH(Vertex, I) = (Vertex == VerticesOnEdgeH(Edge, 0)) ? -1.0 : 1.0;
      }
}

// HorzMesh::setMasks (reference HorzMesh.cpp:581-602): 1 everywhere (sentinel row
// included), 0 on local edges with a cell outside [0, NCellsAll)
void HorzMesh::setMasks() {
   EdgeMask1DH = HostArrayReal(NEdgesSize, 1, 1, 1.0);
   for (int Edge = 0; Edge < NEdgesAll; ++Edge) {
      const I4 Cell1 = CellsOnEdgeH(Edge, 0), Cell2 = CellsOnEdgeH(Edge, 1);
      if (!(Cell1 >= 0 && Cell1 < NCellsAll) || !(Cell2 >= 0 && Cell2 < NCellsAll))
         EdgeMask1DH(Edge) = 0.0;
   }
}

HostArrayReal HorzMesh::edgeMask2D() const {
   HostArrayReal M2(NEdgesSize, NVertLayers, 1, 1.0);
   for (int Edge = 0; Edge < NEdgesSize; ++Edge)
      for (int K = 0; K < NVertLayers; ++K)
         M2(Edge, K) = EdgeMask1DH(Edge);
   return M2;
}

// HorzMesh::setMeshScaling (reference HorzMesh.cpp:607-626): no scaling option only
void HorzMesh::setMeshScaling() {
   MeshScalingDel2H = HostArrayReal(NEdgesSize, 1, 1, 0.0);
   MeshScalingDel4H = HostArrayReal(NEdgesSize, 1, 1, 0.0);
   for (int Edge = 0; Edge < NEdgesAll; ++Edge) {
      MeshScalingDel2H(Edge) = 1.0;
      MeshScalingDel4H(Edge) = 1.0;
   }
}

void HorzMesh::copyToDevice() {
   CellsOnCell    = createDeviceMirrorCopy<I4, 2>("CellsOnCell", CellsOnCellH);
   EdgesOnCell    = createDeviceMirrorCopy<I4, 2>("EdgesOnCell", EdgesOnCellH);
   NEdgesOnCell   = createDeviceMirrorCopy<I4, 1>("NEdgesOnCell", NEdgesOnCellH);
   VerticesOnCell = createDeviceMirrorCopy<I4, 2>("VerticesOnCell", VerticesOnCellH);
   CellsOnEdge    = createDeviceMirrorCopy<I4, 2>("CellsOnEdge", CellsOnEdgeH);
   EdgesOnEdge    = createDeviceMirrorCopy<I4, 2>("EdgesOnEdge", EdgesOnEdgeH);
   NEdgesOnEdge   = createDeviceMirrorCopy<I4, 1>("NEdgesOnEdge", NEdgesOnEdgeH);
   VerticesOnEdge = createDeviceMirrorCopy<I4, 2>("VerticesOnEdge", VerticesOnEdgeH);
   CellsOnVertex  = createDeviceMirrorCopy<I4, 2>("CellsOnVertex", CellsOnVertexH);
   EdgesOnVertex  = createDeviceMirrorCopy<I4, 2>("EdgesOnVertex", EdgesOnVertexH);

   AreaCell          = createDeviceMirrorCopy<Real, 1>("AreaCell", AreaCellH);
   AreaTriangle      = createDeviceMirrorCopy<Real, 1>("AreaTriangle", AreaTriangleH);
   KiteAreasOnVertex = createDeviceMirrorCopy<Real, 2>("KiteAreasOnVertex", KiteAreasOnVertexH);
   DcEdge            = createDeviceMirrorCopy<Real, 1>("DcEdge", DcEdgeH);
   DvEdge            = createDeviceMirrorCopy<Real, 1>("DvEdge", DvEdgeH);
   AngleEdge         = createDeviceMirrorCopy<Real, 1>("AngleEdge", AngleEdgeH);
   WeightsOnEdge     = createDeviceMirrorCopy<Real, 2>("WeightsOnEdge", WeightsOnEdgeH);
   FVertex           = createDeviceMirrorCopy<Real, 1>("FVertex", FVertexH);
   BottomDepth       = createDeviceMirrorCopy<Real, 1>("BottomDepth", BottomDepthH);

   EdgeSignOnCell   = createDeviceMirrorCopy<Real, 2>("EdgeSignOnCell", EdgeSignOnCellH);
   EdgeSignOnVertex = createDeviceMirrorCopy<Real, 2>("EdgeSignOnVertex", EdgeSignOnVertexH);
   MeshScalingDel2  = createDeviceMirrorCopy<Real, 1>("MeshScalingDel2", MeshScalingDel2H);
   MeshScalingDel4  = createDeviceMirrorCopy<Real, 1>("MeshScalingDel4", MeshScalingDel4H);
}

void HorzMesh::setFVertex(const Real *HostValues) {
   for (int V = 0; V < NVerticesSize; ++V)
      FVertexH(V) = HostValues[V];
   if (!HostOnly)
      OMEGA::copyToDevice(FVertex.Ptr, FVertexH.data(), FVertex.bytes());
}

// Level-independent prefixes of the reference's product chains, in the reference's
// left-to-right order (file:line of each chain in the comments).
void HorzMesh::buildCoefficientTables() {
   const int ME = MaxEdges, ME2 = MaxEdges2, VD = VertexDegree;
   const HostArrayReal &Mask1D = EdgeMask1DH;

   HostArrayReal InvA(NCellsSize, 1, 1, 0.0), DvS(NCellsSize, ME, 1, 0.0), DivC(NCellsSize, ME, 1, 0.0),
       KEC(NCellsSize, ME, 1, 0.0), MDvS(NCellsSize, ME, 1, 0.0), D2T(NCellsSize, ME, 1, 0.0),
       Df2(NCellsSize, ME, 1, 0.0), Df4(NCellsSize, ME, 1, 0.0), D2TS(NCellsSize, ME, 1, 0.0),
       Df2S(NCellsSize, ME, 1, 0.0), Df4S(NCellsSize, ME, 1, 0.0);
   HostArrayI4 COEOC(NCellsSize, ME, 2, NCellsAll);
   HostArrayI4 NbrF(NCellsSize, ME, 1, NCellsAll);
   for (int C = 0; C < NCellsAll; ++C) {
      const Real InvAreaCell = 1. / AreaCellH(C);
      InvA(C)                = InvAreaCell;
      for (int J = 0; J < NEdgesOnCellH(C); ++J) {
         const int E     = EdgesOnCellH(C, J);
         const Real Sign = EdgeSignOnCellH(C, J);
         const Real Dv = DvEdgeH(E), Dc = DcEdgeH(E), M = Mask1D(E);
         DvS(C, J)  = Dv * Sign;                 // TendencyTerms.h:48  DvEdge*EdgeSignOnCell*...
         DivC(C, J) = Dv * InvAreaCell * Sign;   // KineticAuxVars.h:38, VelocityDel2AuxVars.h:58
         const Real AreaEdge = 0.5 * Dv * Dc;    // KineticAuxVars.h:32
         KEC(C, J)  = AreaEdge * 0.5 * InvAreaCell; // KineticAuxVars.h:35
         MDvS(C, J) = M * Dv * Sign;             // TendencyTerms.h:363-364
         const Real DvDcEdge = Dv / Dc;          // TracerAuxVars.h:76
         D2T(C, J)           = M * Sign * DvDcEdge; // TracerAuxVars.h:81-82
         const Real RTemp2   = MeshScalingDel2H(E) * Dv / Dc; // TendencyTerms.h:410-411
         Df2(C, J)           = M * Sign * RTemp2;             // TendencyTerms.h:418-419
         const Real RTemp4   = MeshScalingDel4H(E) * Dv / Dc; // TendencyTerms.h:464-465
         Df4(C, J)           = M * Sign * RTemp4;             // TendencyTerms.h:472-473
         COEOC.V[((size_t)C * ME + J) * 2 + 0] = CellsOnEdgeH(E, 0);
         COEOC.V[((size_t)C * ME + J) * 2 + 1] = CellsOnEdgeH(E, 1);
         // neighbour across edge J; bit 30 set when this cell is the edge's first cell
         // (EdgeSignOnCell = -1).  Edges that do not hold this cell at all (the sentinel edge of
         // outer-halo cells) count as "cell is second".
         const bool SelfIsC0 = CellsOnEdgeH(E, 0) == C;
         NbrF(C, J)          = (SelfIsC0 ? CellsOnEdgeH(E, 1) : CellsOnEdgeH(E, 0)) | (SelfIsC0 ? (1 << 30) : 0);
         const Real Orient   = SelfIsC0 ? 1.0 : -1.0; // (T1 - T0) = Orient * (Tneighbour - Tself), exactly
         D2TS(C, J)          = D2T(C, J) * Orient;
         Df2S(C, J)          = Df2(C, J) * Orient;
         Df4S(C, J)          = Df4(C, J) * Orient;
      }
   }
   HostArrayReal KiteC(NVerticesSize, VD, 1, 0.0), VortC(NVerticesSize, VD, 1, 0.0);
   for (int V = 0; V < NVerticesAll; ++V) {
      const Real InvAreaTriangle = 1. / AreaTriangleH(V);
      for (int J = 0; J < VD; ++J) {
         const int E = EdgesOnVertexH(V, J);
         KiteC(V, J) = InvAreaTriangle * KiteAreasOnVertexH(V, J);              // VorticityAuxVars.h:41-42
         VortC(V, J) = InvAreaTriangle * DcEdgeH(E) * EdgeSignOnVertexH(V, J);  // VorticityAuxVars.h:44-45
      }
   }
   HostKiteC = KiteC, HostVortC = VortC;
   HostArrayReal IDc(NEdgesSize, 1, 1, 0.0), IDv(NEdgesSize, 1, 1, 0.0), IDv2(NEdgesSize, 1, 1, 0.0);
   HostArrayI4 PVS(NEdgesSize, ME2, 4, 0);
   for (int E = 0; E < NEdgesSize; ++E)
      for (int J = 0; J < ME2; ++J) {
         int *P = &PVS.V[((size_t)E * ME2 + J) * 4];
         P[0] = P[1] = NCellsAll;
         P[2] = P[3] = NVerticesAll;
      }
   for (int E = 0; E < NEdgesAll; ++E) {
      IDc(E)  = 1. / DcEdgeH(E);                                   // TendencyTerms.h:133
      IDv(E)  = 1. / DvEdgeH(E);                                   // TendencyTerms.h:207
      IDv2(E) = 1. / std::max(DvEdgeH(E), 0.25 * DcEdgeH(E));      // VelocityDel2AuxVars.h:32-33
      for (int J = 0; J < NEdgesOnEdgeH(E); ++J) {
         const int JE = EdgesOnEdgeH(E, J);
         int *P       = &PVS.V[((size_t)E * ME2 + J) * 4];
         P[0]         = CellsOnEdgeH(JE, 0);
         P[1]         = CellsOnEdgeH(JE, 1);
         P[2]         = VerticesOnEdgeH(JE, 0);
         P[3]         = VerticesOnEdgeH(JE, 1);
      }
   }

   // ---- chain form of the PV stencil (see HorzMesh.h) ----
   const int MEm1 = ME - 1;
   HostArrayI4 ChV(NEdgesSize, 2, ME, NVerticesAll), ChF(NEdgesSize, 2, MEm1, NCellsAll),
       ChE(NEdgesSize, 2, MEm1, NEdgesAll);
   HostArrayReal ChW(NEdgesSize, 2, MEm1, 0.0);
   bool ChainOK = true;
   auto SharedVertex = [&](int E1, int E2) {
      for (int A = 0; A < 2; ++A)
         for (int B = 0; B < 2; ++B)
            if (VerticesOnEdgeH(E1, A) == VerticesOnEdgeH(E2, B) && VerticesOnEdgeH(E1, A) < NVerticesAll)
               return (int)VerticesOnEdgeH(E1, A);
      return -1;
   };
   for (int E = 0; E < NEdgesAll && ChainOK; ++E) {
      // an edge with a missing cell (coast line, outermost halo rim) has EdgeMask 0: its PV term is
      // mask * (finite sum) = 0 whatever the stencil, so its tables stay at the zero-weight padding
      if (Mask1D(E) == 0.0)
         continue;
      int Pos = 0; // running position in EdgesOnEdge(E, .)
      for (int Sd = 0; Sd < 2; ++Sd) {
         const int Cs = CellsOnEdgeH(E, Sd);
         if (Cs < 0 || Cs >= NCellsAll)
            continue; // no cell on this side: no stencil entries
         const int N = NEdgesOnCellH(Cs);
         int P0 = -1;
         for (int J = 0; J < N; ++J)
            if (EdgesOnCellH(Cs, J) == E)
               P0 = J;
         if (P0 < 0 || N > ME) {
            ChainOK = false;
            break;
         }
         int Prev = E;
         for (int Kk = 1; Kk < N; ++Kk) {
            // the next entries of EdgesOnEdge(E, :) are the other edges of this side's cell, each sharing a vertex with its
            // predecessor (MPAS writes them walking around the cell).  The chain is taken from EdgesOnEdge itself, not
            // from the cell's EdgesOnCell list: a cell whose own list is not in ring order then only loses its own ring
            // tables (buildCellPV compares the two and marks it, MeshView::BadCells), not the mesh its chain form.
            if (Pos >= ME2) {
               ChainOK = false;
               break;
            }
            const int Ep = EdgesOnEdgeH(E, Pos);
            if (Ep < 0 || Ep >= NEdgesAll || (CellsOnEdgeH(Ep, 0) != Cs && CellsOnEdgeH(Ep, 1) != Cs)) {
               ChainOK = false;
               break;
            }
            const int V = SharedVertex(Prev, Ep);
            if (V < 0) {
               ChainOK = false;
               break;
            }
            const size_t Bv = ((size_t)E * 2 + Sd) * ME, Bm = ((size_t)E * 2 + Sd) * MEm1;
            ChV.V[Bv + Kk - 1] = V;
            const bool SideIsC0 = CellsOnEdgeH(Ep, 0) == Cs;
            const int Far       = SideIsC0 ? CellsOnEdgeH(Ep, 1) : CellsOnEdgeH(Ep, 0);
            ChF.V[Bm + Kk - 1]  = Far | (SideIsC0 ? (1 << 30) : 0);
            ChE.V[Bm + Kk - 1]  = Ep;
            ChW.V[Bm + Kk - 1]  = WeightsOnEdgeH(E, Pos);
            if (Kk == N - 1) { // closing vertex: shared by the last edge and E
               const int Vl = SharedVertex(Ep, E);
               if (Vl < 0) {
                  ChainOK = false;
                  break;
               }
               ChV.V[Bv + Kk] = Vl;
            }
            Prev = Ep;
            ++Pos;
         }
         if (!ChainOK)
            break;
      }
      if (ChainOK && Pos != NEdgesOnEdgeH(E))
         ChainOK = false; // EdgesOnEdge is not [other edges of cell 0, other edges of cell 1]
   }
   PVChainVert   = createDeviceMirrorCopy<I4, 3>("PVChainVert", ChV);
   PVChainFar    = createDeviceMirrorCopy<I4, 3>("PVChainFar", ChF);
   PVChainEdge   = createDeviceMirrorCopy<I4, 3>("PVChainEdge", ChE);
   PVChainWeight = createDeviceMirrorCopy<Real, 3>("PVChainWeight", ChW);

   EdgeMask1D        = createDeviceMirrorCopy<Real, 1>("EdgeMask1D", Mask1D);
   InvAreaCell       = createDeviceMirrorCopy<Real, 1>("InvAreaCell", InvA);
   DvSignOnCell      = createDeviceMirrorCopy<Real, 2>("DvSignOnCell", DvS);
   DivCoefOnCell     = createDeviceMirrorCopy<Real, 2>("DivCoefOnCell", DivC);
   KECoefOnCell      = createDeviceMirrorCopy<Real, 2>("KECoefOnCell", KEC);
   MaskDvSignOnCell  = createDeviceMirrorCopy<Real, 2>("MaskDvSignOnCell", MDvS);
   Del2TrCoefOnCell  = createDeviceMirrorCopy<Real, 2>("Del2TrCoefOnCell", D2T);
   Diff2CoefOnCell   = createDeviceMirrorCopy<Real, 2>("Diff2CoefOnCell", Df2);
   Diff4CoefOnCell   = createDeviceMirrorCopy<Real, 2>("Diff4CoefOnCell", Df4);
   CellsOnEdgeOnCell = createDeviceMirrorCopy<I4, 3>("CellsOnEdgeOnCell", COEOC);
   NbrFlagOnCell     = createDeviceMirrorCopy<I4, 2>("NbrFlagOnCell", NbrF);
   Del2TrCoefSOnCell = createDeviceMirrorCopy<Real, 2>("Del2TrCoefSOnCell", D2TS);
   Diff2CoefSOnCell  = createDeviceMirrorCopy<Real, 2>("Diff2CoefSOnCell", Df2S);
   Diff4CoefSOnCell  = createDeviceMirrorCopy<Real, 2>("Diff4CoefSOnCell", Df4S);
   KiteCoefOnVertex  = createDeviceMirrorCopy<Real, 2>("KiteCoefOnVertex", KiteC);
   VortCoefOnVertex  = createDeviceMirrorCopy<Real, 2>("VortCoefOnVertex", VortC);
   InvDcEdge         = createDeviceMirrorCopy<Real, 1>("InvDcEdge", IDc);
   InvDvEdge         = createDeviceMirrorCopy<Real, 1>("InvDvEdge", IDv);
   InvDvEdgeDel2     = createDeviceMirrorCopy<Real, 1>("InvDvEdgeDel2", IDv2);
   PVStencil         = createDeviceMirrorCopy<I4, 3>("PVStencil", PVS);

   MeshView &W = View;
   W.NCellsOwned = NCellsOwned, W.NCellsAll = NCellsAll, W.NCellsSize = NCellsSize;
   W.NEdgesOwned = NEdgesOwned, W.NEdgesAll = NEdgesAll, W.NEdgesSize = NEdgesSize;
   W.NVerticesOwned = NVerticesOwned, W.NVerticesAll = NVerticesAll, W.NVerticesSize = NVerticesSize;
   W.MaxEdges = MaxEdges, W.MaxEdges2 = MaxEdges2, W.VertexDegree = VertexDegree;
   W.NEdgesOnCell = NEdgesOnCell.Ptr, W.EdgesOnCell = EdgesOnCell.Ptr, W.CellsOnCell = CellsOnCell.Ptr;
   W.VerticesOnCell = VerticesOnCell.Ptr, W.CellsOnEdge = CellsOnEdge.Ptr, W.VerticesOnEdge = VerticesOnEdge.Ptr;
   W.NEdgesOnEdge = NEdgesOnEdge.Ptr, W.EdgesOnEdge = EdgesOnEdge.Ptr, W.CellsOnVertex = CellsOnVertex.Ptr;
   W.EdgesOnVertex = EdgesOnVertex.Ptr;
   W.AreaCell = AreaCell.Ptr, W.AreaTriangle = AreaTriangle.Ptr, W.KiteAreasOnVertex = KiteAreasOnVertex.Ptr;
   W.DcEdge = DcEdge.Ptr, W.DvEdge = DvEdge.Ptr, W.AngleEdge = AngleEdge.Ptr, W.WeightsOnEdge = WeightsOnEdge.Ptr;
   W.FVertex = FVertex.Ptr, W.BottomDepth = BottomDepth.Ptr;
   W.EdgeSignOnCell = EdgeSignOnCell.Ptr, W.EdgeSignOnVertex = EdgeSignOnVertex.Ptr;
   W.EdgeMask1D = EdgeMask1D.Ptr, W.MeshScalingDel2 = MeshScalingDel2.Ptr, W.MeshScalingDel4 = MeshScalingDel4.Ptr;
   W.InvAreaCell = InvAreaCell.Ptr, W.DvSignOnCell = DvSignOnCell.Ptr, W.DivCoefOnCell = DivCoefOnCell.Ptr;
   W.KECoefOnCell = KECoefOnCell.Ptr, W.MaskDvSignOnCell = MaskDvSignOnCell.Ptr;
   W.Del2TrCoefOnCell = Del2TrCoefOnCell.Ptr, W.Diff2CoefOnCell = Diff2CoefOnCell.Ptr;
   W.Diff4CoefOnCell = Diff4CoefOnCell.Ptr, W.CellsOnEdgeOnCell = CellsOnEdgeOnCell.Ptr;
   W.NbrFlagOnCell = NbrFlagOnCell.Ptr;
   W.Del2TrCoefSOnCell = Del2TrCoefSOnCell.Ptr, W.Diff2CoefSOnCell = Diff2CoefSOnCell.Ptr;
   W.Diff4CoefSOnCell = Diff4CoefSOnCell.Ptr;
   W.KiteCoefOnVertex = KiteCoefOnVertex.Ptr, W.VortCoefOnVertex = VortCoefOnVertex.Ptr;
   W.InvDcEdge = InvDcEdge.Ptr, W.InvDvEdge = InvDvEdge.Ptr, W.InvDvEdgeDel2 = InvDvEdgeDel2.Ptr;
   W.PVStencil = PVStencil.Ptr;
   W.PVChainOK = ChainOK ? 1 : 0;
   W.PVChainVert = PVChainVert.Ptr, W.PVChainFar = PVChainFar.Ptr, W.PVChainEdge = PVChainEdge.Ptr;
   W.PVChainWeight = PVChainWeight.Ptr;
   HostChV = ChV, HostChF = ChF, HostChE = ChE, HostNbrF = NbrF;
   HostChW = ChW;
   CellBad.assign(NCellsSize, 0);
   // A pass that finds a new bad cell is followed by one that knows it; the set only grows, so the fixed point is
   // reached after at most NCellsAll + 1 passes (in practice two).  The tables published below are those of a pass
   // that found nothing new: a CellBad set the Del2 / CellPV / L1 tables do not reflect is never published.
   for (I8 Pass = 0; Pass <= (I8)NCellsAll + 1; ++Pass) {
      NewBad = false;
      buildDel2Tables();
      buildCellPV();
      buildCellL1Tables();
      if (!NewBad)
         break;
   }
   OMEGA_REQUIRE(!NewBad, "HorzMesh: the set of cells out of ring order did not reach its fixed point");
   publishBadCells();
   buildBandLists((I4)NCellsHaloH.size());
   buildPatchTables();
   buildNarrowTables();
   // test hook: pretend the mesh is not in MPAS ring order, so that every kernel takes its generic form
   if (tuning().ForceGeneric == 1)
      W.PVChainOK = W.CellPVOK = W.CellPVFinalOK = W.Del2RingOK = W.Del2VertOK = W.CellL1OK = 0;
}

// ---- narrow tables (see HorzMesh.h: narrowView) ----
namespace {
// Dst[r][i][j] = Src[r][i][j] for i < W1n, j < W2n (Src is [Rows][W1][W2])
template <class T>
__global__ void narrowColsKernel(T *Dst, const T *Src, size_t Rows, int W1, int W2, int W1n, int W2n) {
   const size_t N = Rows * (size_t)W1n * W2n;
   for (size_t I = (size_t)blockIdx.x * blockDim.x + threadIdx.x; I < N; I += (size_t)gridDim.x * blockDim.x) {
      const size_t R = I / ((size_t)W1n * W2n);
      const int Rem = (int)(I - R * (size_t)W1n * W2n), A = Rem / W2n, B = Rem - A * W2n;
      Dst[I] = Src[(R * W1 + A) * W2 + B];
   }
}
} // namespace

void HorzMesh::buildNarrowTables() {
   MeshView &W = View;
   W.NWideCells = 0, W.WideCells = nullptr;
   const int ME = MaxEdges;
   std::vector<I4> Wide;
   for (int C = 0; C < NCellsAll; ++C)
      if (NEdgesOnCellH(C) == ME)
         Wide.push_back(C);
   WideCells = Array1DI4("WideCells", (int)std::max<size_t>(Wide.size(), 1));
   if (!Wide.empty())
      OMEGA::copyToDevice(WideCells.Ptr, Wide.data(), Wide.size() * sizeof(I4));
   W.NWideCells = (I4)Wide.size(), W.WideCells = WideCells.Ptr;
   HasNarrow = false;
   // worth it (and possible) when the sweeps already run at MaxEdges-1 and every ring table is valid; the kernels are
   // instantiated for 5..8 slots
   if (!(tuning().NarrowTables != 0 && W.DomM1 && ME >= 6 && ME <= 8 && W.CellL1OK && W.CellPVOK && W.CellPVFinalOK &&
         W.Del2RingOK && W.Del2VertOK && W.NBadCells == 0))
      return; // (no wide cell at all -- tables kept at a file's width by the option KeepMaxEdges -- is fine too)
   const int MN = ME - 1;
   Narrow       = View;
   Narrow.MaxEdges = MN; // (MaxEdges2 and every per-edge / per-vertex table stay as they are: not cell-slot tables)
   auto Cut = [&](auto *&Field, int Inner, int InnerNew) {
      using T = std::remove_const_t<std::remove_pointer_t<std::remove_reference_t<decltype(Field)>>>;
      const size_t Rows = (size_t)NCellsSize;
      auto Buf          = std::make_shared<DeviceBuffer>(Rows * MN * InnerNew * sizeof(T));
      const size_t N    = Rows * MN * InnerNew;
      const unsigned Blocks = (unsigned)std::min<size_t>((N + 255) / 256, 65535);
      hipLaunchKernelGGL(narrowColsKernel<T>, dim3(Blocks), dim3(256), 0, nullptr, static_cast<T *>(Buf->Ptr),
                         const_cast<const T *>(Field), Rows, ME, Inner, MN, InnerNew);
      HIP_CHECK(hipGetLastError());
      NarrowBufs.push_back(Buf);
      Field = static_cast<const T *>(Buf->Ptr);
   };
   MeshView &N = Narrow;
   Cut(N.EdgesOnCell, 1, 1), Cut(N.NbrFlagOnCell, 1, 1), Cut(N.KECoefOnCell, 1, 1), Cut(N.DivCoefOnCell, 1, 1);
   Cut(N.DvSignOnCell, 1, 1), Cut(N.Del2TrCoefOnCell, 1, 1), Cut(N.Del2TrCoefSOnCell, 1, 1), Cut(N.MaskDvSignOnCell, 1, 1);
   Cut(N.Diff2CoefOnCell, 1, 1), Cut(N.Diff2CoefSOnCell, 1, 1), Cut(N.Diff4CoefOnCell, 1, 1), Cut(N.Diff4CoefSOnCell, 1, 1);
   Cut(N.SpokeOnCell, 1, 1), Cut(N.VortSelOnCell, 1, 1), Cut(N.VertRingOnCell, 1, 1), Cut(N.RingVertOnCell, 1, 1);
   Cut(N.PVRoleOnCell, 1, 1), Cut(N.RingSignOnCell, 1, 1), Cut(N.InvDcOnCell, 1, 1), Cut(N.Del2GradMaskSOnCell, 1, 1);
   Cut(N.Del2CurlCoefOnCell, 1, 1), Cut(N.CellsOnCell, 1, 1), Cut(N.VerticesOnCell, 1, 1), Cut(N.EdgeSignOnCell, 1, 1);
   Cut(N.KiteCoefOnCell, 3, 3), Cut(N.VortCoefOnCell, 3, 3), Cut(N.CellsOnEdgeOnCell, 2, 2);
   Cut(N.PVWeightOnCell, ME - 1, MN - 1);
   HIP_CHECK(hipDeviceSynchronize());
   // valences as the narrow sweeps see them: MN is "MaxEdges" (the wide view's M1 list), MN-1 its M1 (the wide M2)
   N.NRingCellsM0 = W.NRingCellsM1, N.RingCellsM0 = W.RingCellsM1;
   N.NRingCellsM1 = W.NRingCellsM2, N.RingCellsM1 = W.RingCellsM2;
   N.NRingCellsM2 = 0;
   N.DomM1        = 0;
   N.NWideCells   = 0; // (the list lives in the wide view)
   HasNarrow      = true;
}

void HorzMesh::publishBadCells() {
   MeshView &W = View;
   std::vector<I4> Bad, NRing(NCellsSize, 0);
   for (int C = 0; C < NCellsAll; ++C) {
      NRing[C] = CellBad[C] ? 99 : NEdgesOnCellH(C);
      if (CellBad[C])
         Bad.push_back(C);
   }
   if (!W.CellL1OK) // (the list of vertices the merged level-1 kernel leaves out only exists with its tables)
      Orphans.clear();
   BadCellsD         = Array1DI4("BadCells", (int)std::max<size_t>(Bad.size(), 1));
   OrphanVerticesD   = Array1DI4("OrphanVertices", (int)std::max<size_t>(Orphans.size(), 1));
   NEdgesOnCellRingD = Array1DI4("NEdgesOnCellRing", NCellsSize);
   if (!Bad.empty())
      OMEGA::copyToDevice(BadCellsD.Ptr, Bad.data(), Bad.size() * sizeof(I4));
   if (!Orphans.empty())
      OMEGA::copyToDevice(OrphanVerticesD.Ptr, Orphans.data(), Orphans.size() * sizeof(I4));
   OMEGA::copyToDevice(NEdgesOnCellRingD.Ptr, NRing.data(), NRing.size() * sizeof(I4));
   W.NBadCells = (I4)Bad.size(), W.BadCells = BadCellsD.Ptr;
   W.NOrphanVertices = (I4)Orphans.size(), W.OrphanVertices = OrphanVerticesD.Ptr;
   W.NEdgesOnCellRing = NEdgesOnCellRingD.Ptr;
}

// Tile patches (see HorzMesh.h): rows in order of first appearance (the tile's own cells first).
void HorzMesh::buildPatchTables() {
   MeshView &W         = View;
   const int Tiles[3]  = {8, 16, 32};
   const int NPs[3]    = {24, 48, 96}; // what the kernel's LDS budget holds (FusedKernelsImpl.h: CellPVFinalTracerPatchBody)
   const int ME        = MaxEdges;
   std::vector<I4> Pos(NCellsSize, -1);
   for (int S = 0; S < MeshView::NPatchSizes; ++S) {
      const int T = Tiles[S], NP = NPs[S];
      const int NTiles = (NCellsAll + T - 1) / T;
      std::vector<I4> Rows((size_t)std::max(NTiles, 1) * NP, -1), OK(std::max(NTiles, 1), 0);
      std::vector<unsigned char> Idx((size_t)NCellsSize * 8, 0);
      std::vector<I4> Touched;
      for (int Tl = 0; Tl < NTiles; ++Tl) {
         const int C0 = Tl * T, C1 = std::min(NCellsAll, C0 + T);
         Touched.clear();
         auto Add = [&](I4 R) {
            if (Pos[R] < 0) {
               Pos[R] = (I4)Touched.size();
               Touched.push_back(R);
            }
            return Pos[R];
         };
         for (int Cc = C0; Cc < C1; ++Cc)
            Add(Cc);
         bool Fits = true;
         for (int Cc = C0; Cc < C1; ++Cc) {
            for (int J = 0; J < 7; ++J) {
               const I4 P = J < ME ? Add(HostNbrF(Cc, J) & 0x3fffffff) : 0;
               Idx[(size_t)Cc * 8 + J] = (unsigned char)std::min<I4>(P, 255);
            }
            Idx[(size_t)Cc * 8 + 7] = (unsigned char)Pos[Cc];
         }
         Fits = (int)Touched.size() <= NP && ME <= 7;
         OK[Tl] = Fits ? 1 : 0;
         for (size_t I = 0; I < Touched.size(); ++I) {
            if (Fits)
               Rows[(size_t)Tl * NP + I] = Touched[I];
            Pos[Touched[I]] = -1;
         }
      }
      NPatchTiles[S] = NTiles, NPatchFallback[S] = 0;
      for (int Tl = 0; Tl < NTiles; ++Tl)
         NPatchFallback[S] += OK[Tl] ? 0 : 1;
      PatchRowsD[S] = Array1DI4("PatchRows", (int)Rows.size());
      PatchOKD[S]   = Array1DI4("PatchOK", (int)OK.size());
      PatchIdxD[S]  = Array1DI4("PatchIdx", (int)(Idx.size() / 4));
      OMEGA::copyToDevice(PatchRowsD[S].Ptr, Rows.data(), Rows.size() * sizeof(I4));
      OMEGA::copyToDevice(PatchOKD[S].Ptr, OK.data(), OK.size() * sizeof(I4));
      OMEGA::copyToDevice(PatchIdxD[S].Ptr, Idx.data(), Idx.size());
      W.PatchNP[S] = NP, W.PatchRows[S] = PatchRowsD[S].Ptr, W.PatchIdx[S] = PatchIdxD[S].Ptr, W.PatchOK[S] = PatchOKD[S].Ptr;
   }
}

// Band / interior split of the local cells for overlapping a halo exchange with interior work
// (see HorzMesh.h): breadth-first distance from the halo cells over CellsOnCell.
void HorzMesh::buildBandLists(I4 HaloWidth) {
   MeshView &W = View;
   std::vector<I4> Dist(NCellsAll, -1), Front, Next;
   for (I4 C = NCellsOwned; C < NCellsAll; ++C) {
      Dist[C] = 0;
      Front.push_back(C);
   }
   for (I4 D = 1; D <= HaloWidth + 1 && !Front.empty(); ++D) {
      Next.clear();
      for (I4 C : Front)
         for (int J = 0; J < MaxEdges; ++J) {
            const I4 Nb = CellsOnCellH(C, J);
            if (Nb >= 0 && Nb < NCellsAll && Dist[Nb] < 0) {
               Dist[Nb] = D;
               Next.push_back(Nb);
            }
         }
      Front.swap(Next);
   }
   std::vector<I4> Band, Inter, Send;
   for (I4 C = 0; C < NCellsAll; ++C)
      (Dist[C] >= 0 ? Band : Inter).push_back(C);
   // halo cells that finish an owned edge (see HorzMesh.h: BandSendCells)
   std::vector<char> OnOwnedEdge(NCellsAll, 0);
   for (I4 E = 0; E < NEdgesOwned; ++E)
      for (int J = 0; J < 2; ++J) {
         const I4 C = CellsOnEdgeH(E, J);
         if (C >= NCellsOwned && C < NCellsAll)
            OnOwnedEdge[C] = 1;
      }
   for (I4 C : Band)
      if (C < NCellsOwned || OnOwnedEdge[C])
         Send.push_back(C);
   BandCells     = Array1DI4("BandCells", (int)std::max<size_t>(Band.size(), 1));
   InteriorCells = Array1DI4("InteriorCells", (int)std::max<size_t>(Inter.size(), 1));
   if (!Band.empty())
      OMEGA::copyToDevice(BandCells.Ptr, Band.data(), Band.size() * sizeof(I4));
   if (!Inter.empty())
      OMEGA::copyToDevice(InteriorCells.Ptr, Inter.data(), Inter.size() * sizeof(I4));
   BandSendCells = Array1DI4("BandSendCells", (int)std::max<size_t>(Send.size(), 1));
   if (!Send.empty())
      OMEGA::copyToDevice(BandSendCells.Ptr, Send.data(), Send.size() * sizeof(I4));
   W.NBandCells = (I4)Band.size(), W.NInteriorCells = (I4)Inter.size();
   W.BandCells = BandCells.Ptr, W.InteriorCells = InteriorCells.Ptr;
   W.NBandSendCells = (I4)Send.size(), W.BandSendCells = BandSendCells.Ptr;
}

// Ring form of the velocity-del2 stencils (see HorzMesh.h).  Works for any mesh whose
// EdgesOnCell lists walk around the cell (consecutive slots share a vertex); anything else
// clears the OK flag and the kernels fall back to the per-edge form.
void HorzMesh::buildDel2Tables() {
   const int ME = MaxEdges, VD = VertexDegree;
   MeshView &W = View;
   HostArrayI4 Ring(NCellsSize, ME, 1, NVerticesAll);
   HostArrayReal GradS(NCellsSize, ME, 1, 0.0), IDcC(NCellsSize, ME, 1, 0.0), CurlC(NCellsSize, ME, 1, 0.0);
   bool RingOK = true;
   auto InvDv2 = [&](int E) { return 1. / std::max(DvEdgeH(E), 0.25 * DcEdgeH(E)); }; // VelocityDel2AuxVars.h:32-33
   for (int C = 0; C < NCellsAll; ++C) {
      const int N = NEdgesOnCellH(C);
      // a cell the ring form cannot describe: left out of the ring tables (sentinel vertices, zero coefficients) and marked
      auto Bad = [&]() {
         markBad(C);
         for (int J = 0; J < ME; ++J)
            Ring(C, J) = NVerticesAll, GradS(C, J) = IDcC(C, J) = CurlC(C, J) = 0.0;
      };
      if (CellBad[C]) {
         Bad();
         continue;
      }
      if (N < 3 || N > ME) {
         Bad();
         continue;
      }
      bool CellOK = true;
      for (int J = 0; J < N && CellOK; ++J) {
         const int E = EdgesOnCellH(C, J), En = EdgesOnCellH(C, (J + 1) % N);
         if (E >= NEdgesAll || En >= NEdgesAll) {
            CellOK = false;
            break;
         }
         int Shared = -1;
         for (int A = 0; A < 2; ++A)
            for (int B = 0; B < 2; ++B)
               if (VerticesOnEdgeH(E, A) == VerticesOnEdgeH(En, B) && VerticesOnEdgeH(E, A) < NVerticesAll)
                  Shared = VerticesOnEdgeH(E, A);
         if (Shared < 0)
            CellOK = false;
         Ring(C, J) = Shared;
      }
      if (!CellOK) {
         Bad();
         continue;
      }
      for (int J = N; J < ME; ++J)
         Ring(C, J) = Ring(C, N - 1); // so that slot 0 finds its first vertex at index ME-1
      for (int J = 0; J < N && CellOK; ++J) {
         const int E  = EdgesOnCellH(C, J);
         const int Vb = Ring(C, J), Va = Ring(C, (J + N - 1) % N);
         Real SV = 0.0;
         if (VerticesOnEdgeH(E, 1) == Vb && VerticesOnEdgeH(E, 0) == Va)
            SV = 1.0;
         else if (VerticesOnEdgeH(E, 0) == Vb && VerticesOnEdgeH(E, 1) == Va)
            SV = -1.0;
         else
            CellOK = false;
         const Real SC = CellsOnEdgeH(E, 0) == C ? 1.0 : -1.0;
         if (CellsOnEdgeH(E, 0) != C && CellsOnEdgeH(E, 1) != C)
            CellOK = false;
         GradS(C, J) = EdgeMask1DH(E) * SC;
         IDcC(C, J)  = 1. / DcEdgeH(E);
         CurlC(C, J) = -SV * InvDv2(E);
      }
      if (!CellOK)
         Bad();
   }
   HostArrayI4 NbrV(NVerticesSize, VD, 1, NVerticesAll), Sel(NVerticesSize, VD, 1, 0);
   HostArrayReal MaskV(NVerticesSize, VD, 1, 0.0), IDcV(NVerticesSize, VD, 1, 0.0), CurlV(NVerticesSize, VD, 1, 0.0);
   bool VertOK = VD == 3;
   for (int V = 0; V < NVerticesAll && VertOK; ++V)
      for (int J = 0; J < VD; ++J) {
         const int E = EdgesOnVertexH(V, J);
         if (E >= NEdgesAll) { // no such edge here (outermost halo): the term must vanish through its
                               // coefficient InvAreaTriangle*DcEdge*EdgeSignOnVertex
            if (DcEdgeH(E) * EdgeSignOnVertexH(V, J) != 0.0)
               VertOK = false;
            NbrV(V, J) = V;
            continue;
         }
         Real SV;
         if (VerticesOnEdgeH(E, 0) == V)
            SV = 1.0, NbrV(V, J) = VerticesOnEdgeH(E, 1);
         else if (VerticesOnEdgeH(E, 1) == V)
            SV = -1.0, NbrV(V, J) = VerticesOnEdgeH(E, 0);
         else {
            VertOK = false;
            break;
         }
         int S0 = -1, S1 = -1;
         for (int A = 0; A < VD; ++A) {
            if (S0 < 0 && CellsOnVertexH(V, A) == CellsOnEdgeH(E, 0))
               S0 = A;
            if (S1 < 0 && CellsOnVertexH(V, A) == CellsOnEdgeH(E, 1))
               S1 = A;
         }
         if (S0 < 0 || S1 < 0) {
            VertOK = false;
            break;
         }
         Sel(V, J)   = S0 | (S1 << 2);
         MaskV(V, J) = EdgeMask1DH(E);
         IDcV(V, J)  = 1. / DcEdgeH(E);
         CurlV(V, J) = -SV * InvDv2(E);
      }
   HostVertRing         = Ring;
   VertRingOnCell       = createDeviceMirrorCopy<I4, 2>("VertRingOnCell", Ring);
   Del2GradMaskSOnCell  = createDeviceMirrorCopy<Real, 2>("Del2GradMaskSOnCell", GradS);
   InvDcOnCell          = createDeviceMirrorCopy<Real, 2>("InvDcOnCell", IDcC);
   Del2CurlCoefOnCell   = createDeviceMirrorCopy<Real, 2>("Del2CurlCoefOnCell", CurlC);
   NbrVertOnVertex      = createDeviceMirrorCopy<I4, 2>("NbrVertOnVertex", NbrV);
   Del2SelOnVertex      = createDeviceMirrorCopy<I4, 2>("Del2SelOnVertex", Sel);
   Del2MaskOnVertex     = createDeviceMirrorCopy<Real, 2>("Del2MaskOnVertex", MaskV);
   InvDcOnVertex        = createDeviceMirrorCopy<Real, 2>("InvDcOnVertex", IDcV);
   Del2CurlCoefOnVertex = createDeviceMirrorCopy<Real, 2>("Del2CurlCoefOnVertex", CurlV);
   W.Del2RingOK = RingOK ? 1 : 0, W.Del2VertOK = VertOK ? 1 : 0;
   W.VertRingOnCell = VertRingOnCell.Ptr, W.Del2GradMaskSOnCell = Del2GradMaskSOnCell.Ptr;
   W.InvDcOnCell = InvDcOnCell.Ptr, W.Del2CurlCoefOnCell = Del2CurlCoefOnCell.Ptr;
   W.NbrVertOnVertex = NbrVertOnVertex.Ptr, W.Del2SelOnVertex = Del2SelOnVertex.Ptr;
   W.Del2MaskOnVertex = Del2MaskOnVertex.Ptr, W.InvDcOnVertex = InvDcOnVertex.Ptr;
   W.Del2CurlCoefOnVertex = Del2CurlCoefOnVertex.Ptr;
}

// Vertex quantities from the cell side (see HorzMesh.h: CellL1OK).
void HorzMesh::buildCellL1Tables() {
   const int ME = MaxEdges, VD = VertexDegree;
   MeshView &W = View;
   HostArrayI4 Spoke(NCellsSize, ME, 1, NEdgesAll), Sel(NCellsSize, ME, 1, 0);
   HostArrayReal KC(NCellsSize, ME, 3, 0.0), VC(NCellsSize, ME, 3, 0.0);
   bool OK = W.Del2RingOK != 0 && VD == 3;
   std::vector<I4> Owner(NVerticesAll, -1), OwnerSlot(NVerticesAll, -1), NOwned(NCellsAll, 0);
   for (int C = 0; C < NCellsAll && OK; ++C) {
      const int N = NEdgesOnCellH(C);
      if (CellBad[C])
         continue; // (served by the generic bodies: no cell-side vertex tables, owns no vertex)
      bool CellOK = true;
      for (int R = 0; R < N && CellOK; ++R) {
         const int V  = HostVertRing(C, R);
         const int E0 = EdgesOnCellH(C, R), E1 = EdgesOnCellH(C, (R + 1) % N);
         const int N0 = HostNbrF(C, R) & 0x3fffffff, N1 = HostNbrF(C, (R + 1) % N) & 0x3fffffff;
         // Roles: cells A = this cell, B = across slot R, C = across slot R+1; edges A = slot R, B = slot R+1,
         // C = the spoke.  The vertex kernel adds its three terms in the vertex's slot order, ((0 + t0) + t1) + t2;
         // t0 + t1 commutes, so all the cell side needs is each role's coefficient and WHICH ROLE IS LAST.
         int Sp = NEdgesAll, LastC = -1, LastE = -1;
         bool SeenC[3] = {false, false, false}, SeenE[3] = {false, false, false};
         for (int J = 0; J < 3 && CellOK; ++J) {
            const int Cv = CellsOnVertexH(V, J), Ev = EdgesOnVertexH(V, J);
            int Rc = -1, Re = -1;
            if (Cv == C && !SeenC[0])
               Rc = 0;
            else if (Cv == N0 && !SeenC[1]) // (a missing cell is the sentinel on both sides: the zero row either way)
               Rc = 1;
            else if (Cv == N1 && !SeenC[2])
               Rc = 2;
            if (Ev == E0 && !SeenE[0])
               Re = 0;
            else if (Ev == E1 && !SeenE[1])
               Re = 1;
            else if (!SeenE[2] && (Ev >= NEdgesAll || (Ev != E0 && Ev != E1)))
               Re = 2, Sp = Ev; // the spoke, or no third edge here (sentinel row)
            if (Rc < 0 || Re < 0) {
               CellOK = false;
               break;
            }
            SeenC[Rc] = SeenE[Re] = true;
            KC.V[((size_t)C * ME + R) * 3 + Rc] = HostKiteC(V, J);
            VC.V[((size_t)C * ME + R) * 3 + Re] = HostVortC(V, J);
            if (J == 2)
               LastC = Rc, LastE = Re;
         }
         if (!CellOK)
            break;
         Spoke(C, R) = Sp;
         Sel(C, R)   = LastC | (LastE << 2);
      }
      if (!CellOK) { // this cell's ring does not match its vertices' lists: a bad cell from the next pass on
         markBad(C);
         continue;
      }
      for (int R = 0; R < N; ++R) {
         const int V = HostVertRing(C, R);
         // ownership: the cell with the fewest stores so far among those that see the vertex
         if (Owner[V] < 0 || NOwned[C] < NOwned[Owner[V]] - 1) {
            if (Owner[V] >= 0)
               --NOwned[Owner[V]];
            Owner[V] = C, OwnerSlot[V] = R;
            ++NOwned[C];
         }
      }
   }
   Orphans.clear();
   for (int V = 0; V < NVerticesAll && OK; ++V) {
      if (Owner[V] < 0)
         Orphans.push_back(V); // no good local cell has it in its ring: the vertex kernel computes it (list launch)
      else
         Sel(Owner[V], OwnerSlot[V]) |= 1 << 4;
   }
   SpokeOnCell    = createDeviceMirrorCopy<I4, 2>("SpokeOnCell", Spoke);
   VortSelOnCell  = createDeviceMirrorCopy<I4, 2>("VortSelOnCell", Sel);
   KiteCoefOnCell = createDeviceMirrorCopy<Real, 3>("KiteCoefOnCell", KC);
   VortCoefOnCell = createDeviceMirrorCopy<Real, 3>("VortCoefOnCell", VC);
   W.SpokeOnCell = SpokeOnCell.Ptr, W.VortSelOnCell = VortSelOnCell.Ptr;
   W.KiteCoefOnCell = KiteCoefOnCell.Ptr, W.VortCoefOnCell = VortCoefOnCell.Ptr;
   // the PV tables number the ring the same way (they are only filled for cells that own regular edges)
   for (int C = 0; C < NCellsAll && OK; ++C)
      for (int R = 0; R < NEdgesOnCellH(C); ++R)
         if (!CellBad[C] && HostPVRole(C, R) != 0 && HostPVRing(C, R) != HostVertRing(C, R))
            markBad(C);
   W.CellL1OK = OK && W.CellPVOK ? 1 : 0;
}

// Cell-centric PV tables (see HorzMesh.h).
void HorzMesh::buildCellPV() {
   const int ME = MaxEdges, MEm1 = ME - 1;
   MeshView &W = View;
   HostArrayI4 Ring(NCellsSize, ME, 1, NVerticesAll), Role(NCellsSize, ME, 1, 0), Reg(NEdgesSize, 1, 1, 0);
   HostArrayReal Wt(NCellsSize, ME, MEm1, 0.0);
   bool OK = W.PVChainOK != 0;
   std::vector<I4> Irregular;
   // valences the ring kernels are instantiated for: MaxEdges, MaxEdges-1 and (MaxEdges >= 6) MaxEdges-2
   auto ValenceOK = [&](int N) { return N == ME || N == ME - 1 || (ME >= 6 && N == ME - 2); };
   if (OK) {
      for (int E = 0; E < NEdgesAll; ++E) {
         const int C0 = CellsOnEdgeH(E, 0), C1 = CellsOnEdgeH(E, 1);
         const bool R = EdgeMask1DH(E) != 0.0 && C0 < NCellsAll && C1 < NCellsAll && ValenceOK(NEdgesOnCellH(C0)) &&
                        ValenceOK(NEdgesOnCellH(C1)) && !CellBad[C0] && !CellBad[C1];
         Reg(E) = R ? 1 : 0;
         if (!R)
            Irregular.push_back(E);
      }
      for (int C = 0; C < NCellsAll && OK; ++C) {
         const int N = NEdgesOnCellH(C);
         if (!ValenceOK(N) || CellBad[C])
            continue; // all its edges are irregular
         bool CellOK = true;
         for (int Kk = 0; Kk < N; ++Kk) {
            const int E = EdgesOnCellH(C, Kk);
            if (E >= NEdgesAll || !Reg(E))
               continue;
            const int Sd = CellsOnEdgeH(E, 0) == C ? 0 : 1;
            Role(C, Kk)  = Sd + 1;
            // the chain of (E, Sd) must be the walk around this cell starting after slot Kk
            for (int J = 1; J < N; ++J) {
               if (HostChE.V[((size_t)E * 2 + Sd) * MEm1 + J - 1] != EdgesOnCellH(C, (Kk + J) % N))
                  CellOK = false;
               Wt.V[((size_t)C * ME + Kk) * MEm1 + J - 1] = HostChW.V[((size_t)E * 2 + Sd) * MEm1 + J - 1];
            }
            // vertex between slot Kk and slot Kk+1 = first chain vertex of this edge on this side
            const int V = HostChV.V[((size_t)E * 2 + Sd) * ME + 0];
            if (Ring(C, Kk) != NVerticesAll && Ring(C, Kk) != V)
               CellOK = false;
            Ring(C, Kk) = V;
            // and the remaining chain vertices are the following ring vertices
            for (int J = 1; J < N; ++J) {
               const int Vj = HostChV.V[((size_t)E * 2 + Sd) * ME + J];
               int &Slot    = Ring(C, (Kk + J) % N);
               if (Slot != NVerticesAll && Slot != Vj)
                  CellOK = false;
               Slot = Vj;
            }
         }
         if (!CellOK) { // its edges' chains do not follow the walk around it: its edges are irregular from the next pass on
            markBad(C);
            for (int Kk = 0; Kk < ME; ++Kk)
               Role(C, Kk) = 0, Ring(C, Kk) = NVerticesAll;
         }
      }
   }
   // orientation of each regular cell's edges against its ring (for the fused side-1 + final pass)
   HostArrayReal RSign(NCellsSize, ME, 1, 1.0);
   bool FinalOK = OK;
   for (int C = 0; C < NCellsAll && FinalOK; ++C) {
      const int N = NEdgesOnCellH(C);
      for (int Kk = 0; Kk < N && Kk < ME; ++Kk) {
         if (Role(C, Kk) == 0 || CellBad[C])
            continue;
         const int E = EdgesOnCellH(C, Kk), Vb = Ring(C, Kk), Va = Ring(C, (Kk + N - 1) % N);
         if (VerticesOnEdgeH(E, 1) == Vb && VerticesOnEdgeH(E, 0) == Va)
            RSign(C, Kk) = 1.0;
         else if (VerticesOnEdgeH(E, 0) == Vb && VerticesOnEdgeH(E, 1) == Va)
            RSign(C, Kk) = -1.0;
         else
            markBad(C);
      }
   }
   RingSignOnCell = createDeviceMirrorCopy<Real, 2>("RingSignOnCell", RSign);
   W.RingSignOnCell = RingSignOnCell.Ptr, W.CellPVFinalOK = FinalOK ? 1 : 0;
   HostPVRing = Ring, HostPVRole = Role;
   RingVertOnCell = createDeviceMirrorCopy<I4, 2>("RingVertOnCell", Ring);
   PVRoleOnCell   = createDeviceMirrorCopy<I4, 2>("PVRoleOnCell", Role);
   PVWeightOnCell = createDeviceMirrorCopy<Real, 3>("PVWeightOnCell", Wt);
   EdgeRegular    = createDeviceMirrorCopy<I4, 1>("EdgeRegular", Reg);
   IrregularEdges = Array1DI4("IrregularEdges", (int)std::max<size_t>(Irregular.size(), 1));
   if (!Irregular.empty())
      OMEGA::copyToDevice(IrregularEdges.Ptr, Irregular.data(), Irregular.size() * sizeof(I4));
   std::vector<I4> CellsM0, CellsM1, CellsM2;
   if (OK)
      for (int C = 0; C < NCellsAll; ++C) {
         const int N = NEdgesOnCellH(C);
         bool Any    = false;
         for (int Kk = 0; Kk < ME; ++Kk)
            Any |= Role(C, Kk) != 0;
         if (Any && N == ME)
            CellsM0.push_back(C);
         if (Any && N == ME - 1)
            CellsM1.push_back(C);
         if (Any && N == ME - 2)
            CellsM2.push_back(C);
      }
   RingCellsM0 = Array1DI4("RingCellsM0", (int)std::max<size_t>(CellsM0.size(), 1));
   if (!CellsM0.empty())
      OMEGA::copyToDevice(RingCellsM0.Ptr, CellsM0.data(), CellsM0.size() * sizeof(I4));
   RingCellsM1 = Array1DI4("RingCellsM1", (int)std::max<size_t>(CellsM1.size(), 1));
   RingCellsM2 = Array1DI4("RingCellsM2", (int)std::max<size_t>(CellsM2.size(), 1));
   if (!CellsM1.empty())
      OMEGA::copyToDevice(RingCellsM1.Ptr, CellsM1.data(), CellsM1.size() * sizeof(I4));
   if (!CellsM2.empty())
      OMEGA::copyToDevice(RingCellsM2.Ptr, CellsM2.data(), CellsM2.size() * sizeof(I4));
   W.NRingCellsM0 = (I4)CellsM0.size(), W.NRingCellsM1 = (I4)CellsM1.size(), W.NRingCellsM2 = (I4)CellsM2.size();
   W.RingCellsM0 = RingCellsM0.Ptr, W.RingCellsM1 = RingCellsM1.Ptr, W.RingCellsM2 = RingCellsM2.Ptr;
   // (the kernels' full sweeps are instantiated for the valence most cells have: MaxEdges or MaxEdges-1)
   W.DomM1 = (ME >= 6 && CellsM1.size() > CellsM0.size()) ? 1 : 0;
   W.CellPVOK = OK ? 1 : 0, W.NIrregularEdges = (I4)Irregular.size();
   W.NIrregularOwned = (I4)(std::lower_bound(Irregular.begin(), Irregular.end(), NEdgesOwned) - Irregular.begin());
   // (Decomp: NEdgesHalo(i) = edges of the cells through halo layer i; the numbering is ascending in the layers)
   const I4 InnerEdges = NEdgesHaloH.size() >= 3 ? NEdgesHaloH(2) : NEdgesAll;
   W.NIrregularInner   = (I4)(std::lower_bound(Irregular.begin(), Irregular.end(), InnerEdges) - Irregular.begin());
   W.RingVertOnCell = RingVertOnCell.Ptr, W.PVRoleOnCell = PVRoleOnCell.Ptr, W.PVWeightOnCell = PVWeightOnCell.Ptr;
   W.EdgeRegular = EdgeRegular.Ptr, W.IrregularEdges = IrregularEdges.Ptr;
}

} // namespace OMEGA
