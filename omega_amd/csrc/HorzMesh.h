// HorzMesh.h -- local horizontal mesh of one rank: connectivity from Decomp, geometry
// from the global mesh arrays, the derived sign / mask / scaling arrays and their
// device mirrors.  Public member names follow the reference
// (components/omega/src/ocn/HorzMesh.h:100-265); derived arrays follow
// components/omega/src/ocn/HorzMesh.cpp:527-626.
//
// MI355X-first additions: `EdgeMask1D` (the reference's EdgeMask(E,K) is level
// independent, HorzMesh.cpp:589-598) and per-(cell,j) / per-(vertex,j) / per-(edge,j)
// coefficient tables that hold the level-independent PREFIX of each reference product
// chain, evaluated on the host in the same left-to-right order, so the device kernels
// reproduce the reference arithmetic bit for bit while staging one coefficient per
// neighbour through LDS.
#ifndef OMEGA_AMD_HORZMESH_H
#define OMEGA_AMD_HORZMESH_H

#include "Base.h"
#include "Decomp.h"

namespace OMEGA {

/// Plain device-pointer view of the mesh handed to kernels by value.
struct MeshView {
   I4 NCellsOwned, NCellsAll, NCellsSize, NEdgesOwned, NEdgesAll, NEdgesSize;
   I4 NVerticesOwned, NVerticesAll, NVerticesSize, MaxEdges, MaxEdges2, VertexDegree;
   const I4 *NEdgesOnCell, *EdgesOnCell, *CellsOnCell, *VerticesOnCell;
   const I4 *CellsOnEdge, *VerticesOnEdge, *NEdgesOnEdge, *EdgesOnEdge;
   const I4 *CellsOnVertex, *EdgesOnVertex;
   const Real *AreaCell, *AreaTriangle, *KiteAreasOnVertex, *DcEdge, *DvEdge, *AngleEdge;
   const Real *WeightsOnEdge, *FVertex, *BottomDepth;
   const Real *EdgeSignOnCell, *EdgeSignOnVertex, *EdgeMask1D;
   const Real *MeshScalingDel2, *MeshScalingDel4;
   // ---- coefficient tables (see HorzMesh::buildCoefficientTables) ----
   const Real *InvAreaCell;        // [C]      1/AreaCell
   const Real *DvSignOnCell;       // [C][ME]  DvEdge*EdgeSignOnCell
   const Real *DivCoefOnCell;      // [C][ME]  DvEdge*InvAreaCell*EdgeSignOnCell
   const Real *KECoefOnCell;       // [C][ME]  (0.5*DvEdge*DcEdge)*0.5*InvAreaCell
   const Real *MaskDvSignOnCell;   // [C][ME]  EdgeMask*DvEdge*EdgeSignOnCell
   const Real *Del2TrCoefOnCell;   // [C][ME]  EdgeMask*EdgeSignOnCell*(DvEdge/DcEdge)
   const Real *Diff2CoefOnCell;    // [C][ME]  EdgeMask*EdgeSignOnCell*(MeshScalingDel2*DvEdge/DcEdge)
   const Real *Diff4CoefOnCell;    // [C][ME]  EdgeMask*EdgeSignOnCell*(MeshScalingDel4*DvEdge/DcEdge)
   // the same three with the orientation folded in (x +1 if this cell is the edge's first cell,
   // else -1), to be used with (neighbour - self) differences: coef*(T1-T0) == coefS*(Tn-Ts) exactly
   const Real *Del2TrCoefSOnCell, *Diff2CoefSOnCell, *Diff4CoefSOnCell; // [C][ME]
   const I4 *CellsOnEdgeOnCell;    // [C][ME][2] CellsOnEdge(EdgesOnCell(c,j), 0..1)
   const I4 *NbrFlagOnCell;        // [C][ME]  cell across edge j, | (1<<30) if this cell is CellsOnEdge(e,0)
   const Real *KiteCoefOnVertex;   // [V][VD]  InvAreaTriangle*KiteAreasOnVertex
   const Real *VortCoefOnVertex;   // [V][VD]  InvAreaTriangle*DcEdge*EdgeSignOnVertex
   const Real *InvDcEdge;          // [E]      1/DcEdge
   const Real *InvDvEdge;          // [E]      1/DvEdge
   const Real *InvDvEdgeDel2;      // [E]      1/max(DvEdge, 0.25*DcEdge)
   // ---- ring form of the velocity-del2 stencils (HorzMesh::buildDel2Tables) ----
   // Cell c, edge slot j: the del2 of the edge needs Div at the cell across (and c itself) and
   // RelVort at the edge's two end vertices, which are ring vertices j-1 and j of the cell, so a
   // thread gathers ME+1 Div rows and ME RelVort rows instead of 4*ME.  Orientation lives in the
   // coefficients (x -1 commutes with rounding): Mask*GradDiv == Del2GradMaskSOnCell*((Dn-Ds)*InvDc).
   I4 Del2RingOK, Del2VertOK;
   const I4 *VertRingOnCell;          // [C][ME] vertex shared by edge slots j and j+1 (cyclic in NEdgesOnCell)
   const Real *Del2GradMaskSOnCell;   // [C][ME] EdgeMask * (+1 if c is CellsOnEdge(e,0) else -1)
   const Real *InvDcOnCell;           // [C][ME] 1/DcEdge
   const Real *Del2CurlCoefOnCell;    // [C][ME] -(+1 if ring vertex j is VerticesOnEdge(e,1) else -1) * InvDvEdgeDel2
   // Vertex v, edge slot j (VertexDegree 3): Div at CellsOnVertex(v,0..2), RelVort at v and at the
   // vertex across each edge: 7 gathers instead of 12.
   const I4 *NbrVertOnVertex;         // [V][3] other end of edge slot j
   const I4 *Del2SelOnVertex;         // [V][3] slot in CellsOnVertex of CellsOnEdge(e,0) | slot of CellsOnEdge(e,1) << 2
   const Real *Del2MaskOnVertex;      // [V][3] EdgeMask
   const Real *InvDcOnVertex;         // [V][3] 1/DcEdge
   const Real *Del2CurlCoefOnVertex;  // [V][3] -(+1 if the other end is VerticesOnEdge(e,1) else -1) * InvDvEdgeDel2
   const I4 *PVStencil;            // [E][ME2][4] CellsOnEdge / VerticesOnEdge of EdgesOnEdge(e,j)
   // Chain form of the PotentialVortHAdvOnEdge stencil (valid when PVChainOK): side s = 0,1 is
   // cell CellsOnEdge(e,s); its other edges e'_1..e'_{n-1} in EdgesOnEdge order are
   // PVChainEdge[e][s][j-1] with weight PVChainWeight[e][s][j-1]; e'_j has end vertices
   // PVChainVert[e][s][j-1], PVChainVert[e][s][j] and far cell PVChainFar[e][s][j-1]
   // (| 1<<30 when the side cell is CellsOnEdge(e'_j, 0)).  Padding: zero weight, sentinel rows.
   I4 PVChainOK;
   const I4 *PVChainVert;          // [E][2][ME]
   const I4 *PVChainFar;           // [E][2][ME-1]
   const I4 *PVChainEdge;          // [E][2][ME-1]
   const Real *PVChainWeight;      // [E][2][ME-1]
   // ---- cell-centric form of the PV stencil (valid when CellPVOK; HorzMesh::buildCellPV) ----
   // A "regular" edge has EdgeMask 1 and two local cells with MaxEdges-2 .. MaxEdges edges (the valences
   // the ring kernels are instantiated for: mesh files carry maxEdges = 7 with 5-, 6- and 7-gons).  For such an edge the side-s
   // part of the PotentialVortHAdvOnEdge sum only needs data on the ring of cell s (its edges,
   // neighbour cells and vertices), so one thread per cell can produce the side sums of all its
   // edges from 4*MaxEdges+1 gathers.  Side 0 sums start from zero, side 1 sums continue from the
   // stored side 0 value: the additions happen in the reference's order.
   I4 CellPVOK, CellPVFinalOK, NIrregularEdges;
   I4 NIrregularOwned;             // how many of them are owned edges (the list is ascending: they come first)
   I4 NIrregularInner;             // ... are edges of cells through halo layer 2 (= all of them below HaloWidth 3)
   const I4 *RingVertOnCell;       // [C][ME] vertex shared by edge slots k and k+1 (cyclic)
   const Real *RingSignOnCell;     // [C][ME] +1 if ring vertex k is VerticesOnEdge(e_k,1) (and k-1 is (e_k,0)), -1 if reversed
   const I4 *PVRoleOnCell;         // [C][ME] 0 none, 1 this cell is cell 0 of a regular edge, 2 cell 1
   const Real *PVWeightOnCell;     // [C][ME][ME-1] WeightsOnEdge of edge slot k, this cell's side, in walk order
   const I4 *EdgeRegular;          // [E] 1 regular, 0 handled by the edge-centric kernel
   const I4 *IrregularEdges;       // [NIrregularEdges]
   // cells of the rarer valences that own regular edges: the ring kernels run once more over each list
   I4 NRingCellsM0, NRingCellsM1, NRingCellsM2;  // cells of valence MaxEdges, MaxEdges-1, MaxEdges-2 (with regular edges)
   const I4 *RingCellsM0, *RingCellsM1, *RingCellsM2;
   I4 DomM1; ///< 1: most cells have MaxEdges-1 edges (a mesh of hexagons with a few heptagons): the full sweeps take that valence
   // ---- narrow tables (HorzMesh::buildNarrowTables): the cells of the widest valence as a list ----
   I4 NWideCells;          ///< cells with MaxEdges edges (wide view), 0 in the narrow view
   const I4 *WideCells;
   // ---- vertex quantities evaluated from the cell side (valid when CellL1OK; HorzMesh::buildCellL1Tables) ----
   // Ring vertex r of cell c (shared by edge slots r and r+1) touches the cells {c, across slot r, across slot r+1}
   // and the edges {slot r, slot r+1, "spoke" = the edge between the two neighbours}.  A thread that already holds
   // h at the cell and its neighbours and u on its edges therefore only gathers the spokes to evaluate
   // VorticityAuxVars::computeVarsOnVertex at all its ring vertices -- with the vertex's own coefficients and in
   // the vertex's own slot order (selectors below), so the bits equal the vertex kernel's.  Every vertex is
   // stored by exactly one of its cells (own bit).
   I4 CellL1OK;
   const I4 *SpokeOnCell;          // [C][ME] third edge of ring vertex r (sentinel row if it has none here)
   const I4 *VortSelOnCell;        // [C][ME] bits 0-1: which of the cells {0 self, 1 across slot r, 2 across slot r+1} sits
                                   //   in the vertex's LAST slot (its term is added last; the first two commute);
                                   //   bits 2-3: the same for the edges {0 slot r, 1 slot r+1, 2 spoke}; bit 4: this
                                   //   cell stores vertex v
   const Real *KiteCoefOnCell;     // [C][ME][3] KiteCoefOnVertex of the vertex's slot holding {self, across r, across r+1}
   const Real *VortCoefOnCell;     // [C][ME][3] VortCoefOnVertex of the vertex's slot holding {slot r, slot r+1, spoke}
   // ---- overlap of halo exchanges with interior work (HorzMesh::buildBandLists) ----
   // BandCells: every halo cell and every owned cell within HaloWidth+1 cells of one (a superset of the cells
   // that own anything a neighbour receives); InteriorCells: the other owned cells.  Ascending order.
   I4 NBandCells, NInteriorCells;
   const I4 *BandCells, *InteriorCells;
   // BandSendCells: BandCells without the halo cells that finish nothing a neighbour receives -- the owned band cells
   // plus the halo cells on an owned edge (the cell-centric velocity kernels finish an edge in the thread of its
   // second cell).  A stage whose output is exchanged right away needs its last dependency level on these and on the
   // interior only: every other local value is about to be overwritten by the exchange.
   I4 NBandSendCells;
   const I4 *BandSendCells;
   // ---- tile patches (HorzMesh::buildPatchTables) ----
   // For the tile sizes the cell sweeps use (8, 16, 32 consecutive cells), tile t: PatchRows[s][t][0 .. PatchNP[s]) = the
   // distinct cell rows its cells and their neighbours touch (-1 = unused slot), PatchIdx[s][c][j] (bytes: j < 7 the cell
   // across edge slot j, j = 7 the cell itself) = position of that row in the tile's list, PatchOK[s][t] = the list fits.
   // The level-3 kernel stages each tracer's rows of a tile once per workgroup into LDS (straight from the buffer, no
   // registers) instead of gathering them per thread: a row is loaded once, not by up to 7 threads, and the next
   // tracer's rows are in flight while this one is computed.
   // ---- cells outside the ring tables (HorzMesh::BadCells) ----
   // A cell whose EdgesOnCell list does not walk around the cell (consecutive slots sharing a vertex), or whose edges'
   // EdgesOnEdge chains do not follow that walk, cannot use the ring-form tables.  The reference does not care about the
   // order (components/omega/src/base/Decomp.cpp:2030-2064 only compacts the lists), so such a cell must not cost the
   // mesh its fast kernels: it is served by the generic per-cell bodies over the list BadCells -- its edges are irregular
   // (edge-centric chain kernel), the ring-form sweeps skip it (NEdgesOnCellRing holds 99 for it), the vertices no good
   // cell stores are on OrphanVertices -- and everything else keeps the fast paths.
   I4 NBadCells, NOrphanVertices;
   const I4 *BadCells, *OrphanVertices;
   const I4 *NEdgesOnCellRing; // [C] NEdgesOnCell, 99 for a bad cell
   static constexpr int NPatchSizes = 3;
   I4 PatchNP[NPatchSizes];
   const I4 *PatchRows[NPatchSizes], *PatchIdx[NPatchSizes], *PatchOK[NPatchSizes];
   static constexpr int patchSlot(int Tile) { return Tile == 8 ? 0 : (Tile == 16 ? 1 : (Tile == 32 ? 2 : -1)); }
};

class HorzMesh : public Registry<HorzMesh> {
 public:
   HorzMesh(const std::string &Name, const Decomp *MeshDecomp, I4 NVertLayers, bool HostOnly = false);
   bool HostOnly; ///< host arrays only (no device mirrors): compute calls are rejected

   std::string MeshName;
   I4 NVertLayers;

   I4 NCellsOwned, NCellsAll, NCellsSize;
   I4 NEdgesOwned, NEdgesAll, NEdgesSize, MaxCellsOnEdge, MaxEdges, MaxEdges2;
   I4 MaxEdgesFile = 0; ///< the mesh file's maxEdges dimension (Decomp::MaxEdges); MaxEdges is the largest valence
                        ///< present on this rank (HorzMesh.cpp: compactMaxEdges)
   I4 NVerticesOwned, NVerticesAll, NVerticesSize, VertexDegree;
   HostArrayI4 NCellsHaloH, NEdgesHaloH, NVerticesHaloH;

   // connectivity (host + device)
   HostArrayI4 CellsOnCellH, EdgesOnCellH, NEdgesOnCellH, VerticesOnCellH, CellsOnEdgeH, EdgesOnEdgeH,
       NEdgesOnEdgeH, VerticesOnEdgeH, CellsOnVertexH, EdgesOnVertexH;
   Array2DI4 CellsOnCell, EdgesOnCell, VerticesOnCell, CellsOnEdge, EdgesOnEdge, VerticesOnEdge, CellsOnVertex,
       EdgesOnVertex;
   Array1DI4 NEdgesOnCell, NEdgesOnEdge;

   // coordinates (host only, as in the reference, plus X/Y device copies)
   HostArrayReal XCellH, YCellH, ZCellH, LonCellH, LatCellH, XEdgeH, YEdgeH, ZEdgeH, LonEdgeH, LatEdgeH, XVertexH,
       YVertexH, ZVertexH, LonVertexH, LatVertexH;

   // measurements, weights, Coriolis, depth
   HostArrayReal AreaCellH, AreaTriangleH, KiteAreasOnVertexH, DvEdgeH, DcEdgeH, AngleEdgeH, WeightsOnEdgeH, FEdgeH,
       FCellH, FVertexH, BottomDepthH;
   Array1DReal AreaCell, AreaTriangle, DvEdge, DcEdge, AngleEdge, FVertex, BottomDepth;
   Array2DReal KiteAreasOnVertex, WeightsOnEdge;

   // derived
   /// EdgeMask: the reference stores EdgeMask(NEdgesSize, NVertLayers) but sets it level-independent
   /// (HorzMesh.cpp:589-598); here it is kept per edge and expanded on request (edgeMask2D()).
   HostArrayReal EdgeSignOnCellH, EdgeSignOnVertexH, EdgeMask1DH, MeshScalingDel2H, MeshScalingDel4H;
   Array2DReal EdgeSignOnCell, EdgeSignOnVertex;
   Array1DReal EdgeMask1D, MeshScalingDel2, MeshScalingDel4;
   HostArrayReal edgeMask2D() const; ///< (NEdgesSize, NVertLayers) as in the reference

   /// Replace FVertex (the reference's tests override it, AuxiliaryVarsTest.cpp:333-338)
   void setFVertex(const Real *HostValues /* NVerticesSize */);

   const MeshView &view() const { return View; }
   /// The same mesh with every per-(cell, slot) table stored MaxEdges-1 wide (rows of the cells with MaxEdges edges are
   /// truncated and must be skipped: view().WideCells lists them), or nullptr.  Built when most cells have MaxEdges-1
   /// edges and every ring table is valid -- a real MPAS mesh: hexagons, 12+ pentagons, a few heptagons.  The fused RHS
   /// then sweeps the 6-wide tables with the 6-slot kernels and runs the heptagons through list launches of the 7-slot
   /// kernels on the wide tables (components/omega/src/base/Decomp.cpp:2043-2064: the reference compacts NEdgesOnCell
   /// per cell, not per mesh).
   const MeshView *narrowView() const { return HasNarrow ? &Narrow : nullptr; }

 private:
   void compactMaxEdges();
   void computeEdgeSign();   // HorzMesh.cpp:527-575
   void setMasks();          // HorzMesh.cpp:581-602
   void setMeshScaling();    // HorzMesh.cpp:607-626
   void copyToDevice();      // HorzMesh.cpp:630-...
   void buildCoefficientTables();

   MeshView View{};
   MeshView Narrow{};
   bool HasNarrow = false;
   void buildNarrowTables();
   std::vector<std::shared_ptr<DeviceBuffer>> NarrowBufs; ///< the narrow copies of the per-(cell, slot) tables
   Array1DI4 WideCells;
   // coefficient tables (device)
   Array1DReal InvAreaCell, InvDcEdge, InvDvEdge, InvDvEdgeDel2;
   Array2DReal DvSignOnCell, DivCoefOnCell, KECoefOnCell, MaskDvSignOnCell, Del2TrCoefOnCell, Diff2CoefOnCell,
       Diff4CoefOnCell, KiteCoefOnVertex, VortCoefOnVertex, Del2TrCoefSOnCell, Diff2CoefSOnCell, Diff4CoefSOnCell;
   DeviceArray<I4, 3> CellsOnEdgeOnCell, PVStencil;
   Array2DReal RingSignOnCell;
   Array1DI4 RingCellsM0, RingCellsM1, RingCellsM2, BandCells, InteriorCells, BandSendCells;
   void buildBandLists(I4 HaloWidth);
   void buildPatchTables();
   /// cells the ring tables cannot describe (see MeshView::BadCells); the three table builders mark them and are run
   /// again until no new one turns up
   std::vector<char> CellBad;
   bool NewBad = false;
   void markBad(int C) {
      if (!CellBad[C])
         CellBad[C] = 1, NewBad = true;
   }
   void publishBadCells();
   Array1DI4 BadCellsD, OrphanVerticesD, NEdgesOnCellRingD;
   std::vector<I4> Orphans;
   /// tiles per patch size (8, 16, 32 cells) and how many of them do NOT fit the patch (their cells take the per-thread
   /// gathers inside the same launch): diagnostics, omg_mesh_get_int "NPatchTiles16" / "NPatchFallback16" ...
 public:
   I4 NPatchTiles[MeshView::NPatchSizes] = {0, 0, 0}, NPatchFallback[MeshView::NPatchSizes] = {0, 0, 0};
 private:
   Array1DI4 PatchRowsD[MeshView::NPatchSizes], PatchIdxD[MeshView::NPatchSizes], PatchOKD[MeshView::NPatchSizes];
   Array2DI4 NbrFlagOnCell, VertRingOnCell, NbrVertOnVertex, Del2SelOnVertex;
   Array2DReal Del2GradMaskSOnCell, InvDcOnCell, Del2CurlCoefOnCell, Del2MaskOnVertex, InvDcOnVertex, Del2CurlCoefOnVertex;
   DeviceArray<I4, 3> PVChainVert, PVChainFar, PVChainEdge;
   Array3DReal PVChainWeight;
   void buildCellPV();
   void buildDel2Tables();
   void buildCellL1Tables();
   HostArrayI4 HostVertRing, HostPVRing, HostPVRole;
   HostArrayReal HostKiteC, HostVortC;
   Array2DI4 SpokeOnCell, VortSelOnCell;
   Array3DReal KiteCoefOnCell, VortCoefOnCell;
   Array2DI4 RingVertOnCell, PVRoleOnCell;
   Array3DReal PVWeightOnCell;
   Array1DI4 EdgeRegular, IrregularEdges;
   HostArrayI4 HostChV, HostChF, HostChE, HostNbrF;
   HostArrayReal HostChW;
};

} // namespace OMEGA
#endif
