// Tuning.h -- the library's test switches, set through an explicit API (omg_set_option), never through the environment:
// a backend that drops into components/omega must not change its kernel structure because a job script exports a
// variable.  The defaults are what production runs.  Every switch forces a structure that SOME mesh class reaches in
// production without the switch -- named next to it -- so that the tests can drive generated meshes (which never take
// those branches by themselves) through them; there is no switch that selects a structure no mesh takes, and no
// measurement probe: experiments live in variant builds (Makefile: VARIANT= EXTRA=-D...), their results in
// profiles/EXPERIMENTS.md.
#ifndef OMEGA_AMD_TUNING_H
#define OMEGA_AMD_TUNING_H

#include <string>

namespace OMEGA {

struct TuningOptions {
   // ---- kernel structure of the fused RHS (FusedKernelsImpl.h: launchFusedT)
   int MergeL1 = 1;     ///< 0: vertex pass, level-1 cell kernel and side-0 PV sums as three kernels -- what a mesh without the
                        ///< cell-side vertex tables takes (MeshView::CellL1OK false: a cell whose vertex ring is not closed in
                        ///< MPAS order) and what a non-default term set takes on 8-wide tables
   int Pair = 1;        ///< 0: independent sweeps as separate launches -- what a mesh without the ring-form del2 tables takes
                        ///< (MeshView::Del2RingOK / Del2VertOK false) and what NT = 0 takes at level 3
   int TracerPatch = 1; ///< 0: level 3 of the plain RHS gathers the tracers' neighbour rows per thread instead of through LDS
                        ///< tile patches -- what NT < 4, K odd and tiles touching more than 48 distinct rows take
   // ---- the N > 1 shortcuts of the RK4 stages (cross-checks: the same bits with and without, tests/test_00_multirank_gpu.py)
   int SendBand = 1;     ///< 0: the band launches of an exchanged stage keep the halo cells whose results the exchange replaces
   int BandOnComm = 1;   ///< 0: the band launches stay on the compute stream
   int ShrinkSweeps = 1; ///< 0: every stage sweeps every local cell (what HaloWidth < 4 takes in stages 0 and 2 anyway)
   // ---- mesh tables (read when a HorzMesh is constructed)
   int ForceGeneric = 0; ///< 1: every ring-table flag cleared -- the generic kernels (FusedEdgeBody, FusedDel2CellBody,
                         ///< FusedDel2VertexBody, edge-centric chain) of a mesh whose EdgesOnCell / EdgesOnEdge lists are not
                         ///< in MPAS ring order
   int KeepMaxEdges = 0; ///< 1: the file's maxEdges as the table width -- what a mesh with real 8-valent cells takes
   int NarrowTables = 1; ///< 0: one set of cell tables, MaxEdges wide -- what a hexagon mesh with heptagons takes when a ring
                         ///< table of it is not valid
   // ---- HIP-graph replay: -1 = as each object's UseGraphs says, 0 = never, 1 = default on
   int Graphs = -1;
};

TuningOptions &tuning();
/// bumped by every setTuningOption: captured launch sequences (GraphCache keys) are only replayed under the options
/// they were captured with
unsigned long long tuningGeneration();
/// false if there is no option of that name
bool setTuningOption(const std::string &Name, int Value);
bool getTuningOption(const std::string &Name, int &Value);

} // namespace OMEGA
#endif
