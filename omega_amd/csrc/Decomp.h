// Decomp.h -- domain decomposition of the global MPAS mesh for one rank.
//
// Same contract as the reference's Decomp (components/omega/src/base/Decomp.h:189-260;
// construction flow components/omega/src/base/Decomp.cpp:444-745): cells are
// partitioned, `HaloWidth` cell halo layers are added (each layer sorted by global
// ID), edges / vertices are owned by the first valid cell of CellsOnEdge /
// CellsOnVertex, local order = owned, halo-1, halo-2, ..., then one sentinel slot
// (index NXxAll) that every missing / non-local neighbour maps to.
//
// MI355X-first differences (design, not numerics):
//  * one process per GPU holds the whole global connectivity on the host, so the
//    reference's round-robin MPI_Bcast choreography disappears: every rank derives
//    any rank's numbering deterministically (that is also how Halo gets its send
//    lists without an index exchange);
//  * METIS is not available: the built-in partitioner is recursive coordinate
//    bisection (RCB); a caller-supplied cell->task vector (e.g. a METIS
//    graph.info.part.N file) is honoured as is.
#ifndef OMEGA_AMD_DECOMP_H
#define OMEGA_AMD_DECOMP_H

#include "Base.h"

namespace OMEGA {

/// Global mesh as a mesh-file reader hands it over: host pointers, 0-based indices,
/// -1 = missing (MPAS files: 1-based, 0 = missing; Decomp.cpp:108-395).
struct GlobalMeshDesc {
   I4 NCells = 0, NEdges = 0, NVertices = 0, MaxEdges = 0, VertexDegree = 3;
   const I4 *CellsOnCell = nullptr, *EdgesOnCell = nullptr, *VerticesOnCell = nullptr;
   const I4 *CellsOnEdge = nullptr, *VerticesOnEdge = nullptr, *EdgesOnEdge = nullptr;
   const I4 *CellsOnVertex = nullptr, *EdgesOnVertex = nullptr;
   // geometry (HorzMesh reads these; components/omega/src/ocn/HorzMesh.cpp:424-523)
   const R8 *XCell = nullptr, *YCell = nullptr, *ZCell = nullptr, *LonCell = nullptr, *LatCell = nullptr;
   const R8 *XEdge = nullptr, *YEdge = nullptr, *ZEdge = nullptr, *LonEdge = nullptr, *LatEdge = nullptr;
   const R8 *XVertex = nullptr, *YVertex = nullptr, *ZVertex = nullptr, *LonVertex = nullptr, *LatVertex = nullptr;
   const R8 *AreaCell = nullptr, *AreaTriangle = nullptr, *KiteAreasOnVertex = nullptr;
   const R8 *DcEdge = nullptr, *DvEdge = nullptr, *AngleEdge = nullptr, *WeightsOnEdge = nullptr;
   const R8 *FCell = nullptr, *FEdge = nullptr, *FVertex = nullptr, *BottomDepth = nullptr;
};

enum PartMethod { PartMethodRCB, PartMethodUser };

/// Order of a rank's cells inside each group (owned, halo layer 1, 2, ...).  GlobalID is the reference's
/// (Decomp.cpp:1000-1080: owned cells in global-id order, every halo layer sorted by global id); Curve orders
/// each group along a space-filling (Morton) curve through the cell centres, so that consecutive local cells --
/// a kernel's tile, an XCD's share of the sweep -- are spatial neighbours whatever numbering the mesh file came
/// with; Hilbert does the same along a Hilbert curve (no jumps: consecutive cells are always adjacent boxes of the
/// quantisation grid).  Edges and vertices follow the cells (order of encounter) in every case.  Results per global
/// id are identical; only the local numbering differs.
/// KdTree orders each group by recursive median bisection along the widest axis of the subset's bounding box, down to
/// leaves of 8 cells (splits on multiples of 32 cells above that): every aligned run of 8 / 16 / 32 local cells -- a
/// kernel's tile -- is a compact, roughly square patch ON THE SURFACE the cells live on.  On a sphere the 3-D Morton and
/// Hilbert curves cut the surface with an axis-aligned grid: their 16-cell runs are 9.7 cell spacings across and touch
/// 41.9 distinct cell rows (cell + neighbours) against 4.9 spacings / 35.0 rows for the k-d order and 4.7 / 34 for a
/// planar 4 x 4 block (QU-sized icosahedral and Fibonacci spheres, 163 842 cells).  The alignment holds for the OWNED
/// group, which starts at local index 0; every halo layer is a group of its own that starts where the previous group
/// ends (not on a multiple of 32), so a kernel's tile in the halo range is a compact patch only up to that offset.
/// The leaves (<= 8 cells) are sorted by (coordinate along the leaf's widest axis, global id): the numbering is a
/// function of the coordinates alone, independent of the standard library's nth_element.
enum class LocalOrder { GlobalID = 0, Curve = 1, Hilbert = 2, KdTree = 3 };

/// Ordered local element lists of one rank (global 0-based ids) with layer bounds.
struct LocalSets {
   std::vector<I4> CellID, EdgeID, VertexID;
   I4 NCellsOwned = 0, NEdgesOwned = 0, NVerticesOwned = 0;
   std::vector<I4> NCellsHalo, NEdgesHalo, NVerticesHalo; ///< HaloWidth entries each
};

class Decomp : public Registry<Decomp> {
 public:
   Decomp(const GlobalMeshDesc &Mesh, I4 NParts, I4 MyTask, I4 HaloWidth,
          const I4 *UserCellTask /* nullable */, LocalOrder Order = LocalOrder::GlobalID);
   /// named form for Decomp::create(Name, ...) (Decomp.h:262-275)
   Decomp(const std::string &Name, const GlobalMeshDesc &Mesh, I4 NParts, I4 MyTask, I4 HaloWidth,
          const I4 *UserCellTask = nullptr, LocalOrder Order = LocalOrder::GlobalID)
       : Decomp(Mesh, NParts, MyTask, HaloWidth, UserCellTask, Order) {
      (void)Name;
   }

   // ---- public data, names as in the reference (host side; "H" arrays) ----
   I4 HaloWidth;
   I4 NumTasks, MyTask;
   LocalOrder Order;

   I4 NCellsGlobal, NCellsOwned, NCellsAll, NCellsSize, MaxEdges;
   HostArrayI4 NCellsHaloH; ///< [HaloWidth] owned+halo count through layer i
   HostArrayI4 CellIDH;     ///< [NCellsSize] 1-based global id (sentinel: NCellsGlobal+1)
   HostArrayI4 CellLocH;    ///< [NCellsSize][2] (task, local address on that task)

   I4 NEdgesGlobal, NEdgesOwned, NEdgesAll, NEdgesSize, MaxCellsOnEdge = 2;
   HostArrayI4 NEdgesHaloH, EdgeIDH, EdgeLocH;

   I4 NVerticesGlobal, NVerticesOwned, NVerticesAll, NVerticesSize, VertexDegree;
   HostArrayI4 NVerticesHaloH, VertexIDH, VertexLocH;

   HostArrayI4 CellsOnCellH, EdgesOnCellH, NEdgesOnCellH, VerticesOnCellH;
   HostArrayI4 CellsOnEdgeH, EdgesOnEdgeH, NEdgesOnEdgeH, VerticesOnEdgeH;
   HostArrayI4 CellsOnVertexH, EdgesOnVertexH;

   // ---- global tables every rank derives identically ----
   std::vector<I4> CellTask;    ///< [NCellsGlobal] owner task of each cell
   std::vector<I4> CellLocAll;  ///< [NCellsGlobal] local address on the owner
   std::vector<I4> EdgeTask, EdgeLocAll, VertexTask, VertexLocAll;

   /// Ordered element lists of any rank (used by Halo to build send lists).
   LocalSets computeLocalSets(I4 Task) const;

   const GlobalMeshDesc &globalMesh() const { return G; }

 private:
   GlobalMeshDesc G;
   std::vector<I4> CellSeq;  ///< cells in the order that numbers them (identity, or along the curve)
   std::vector<I4> CellRank; ///< position of each cell in CellSeq
   std::vector<std::vector<I4>> OwnedSeq; ///< per task: its owned cells in numbering order
   /// k-d order of List[Begin, End) in place (LocalOrder::KdTree)
   void kdOrder(std::vector<I4> &List, size_t Begin, size_t End) const;
   void kdOrderRange(std::vector<I4> &List, size_t Begin, size_t End) const;
   /// what is applied to every group (owned cells of a task, a halo layer) after the sequence order
   void orderGroup(std::vector<I4> &List, size_t Begin, size_t End) const {
      if (Order == LocalOrder::KdTree)
         kdOrder(List, Begin, End);
   }
   void buildCellOrder();
   void partitionRCB();
   void computeOwnership();
   void buildLocalConnectivity(const LocalSets &S);
};

} // namespace OMEGA
#endif
