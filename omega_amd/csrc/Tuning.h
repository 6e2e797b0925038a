// Tuning.h -- measurement / test switches of the library, set through an explicit API (omg_set_option), never through
// the environment: a backend that drops into components/omega must not change its kernel structure because a job
// script exports a variable.  The defaults are what production runs; everything else exists so that tests can force
// the fallback structures a generated mesh would never take, and so that A/B measurements can be made on one build.
// (A build with -DOMEGA_TUNING_ENV additionally initialises the options from OMEGA_<NAME> environment variables --
// not the default build.)
#ifndef OMEGA_AMD_TUNING_H
#define OMEGA_AMD_TUNING_H

#include <string>

namespace OMEGA {

struct TuningOptions {
   // ---- tile geometry (KernelCommon.h: makeGeom); 0 / -1 = the built-in choice
   int W          = 2;  ///< levels per thread (1: scalar accesses everywhere)
   int TX         = 0;  ///< threads along the levels
   int TY         = 0;  ///< elements per workgroup
   int Sweeps     = 1;  ///< tiles per workgroup
   int ChunkSplit = -1; ///< level chunks over gridDim.y (-1: by sweep size)
   int TailSplit  = 1;  ///< spread the last partial round of workgroups over the level chunks
   // ---- kernel structure of the fused RHS (FusedKernels.hip: launchFusedT)
   int EdgeMode  = 0; ///< 1: edge-centric chain kernel instead of the cell-centric PV kernels
   int FuseFinal = 1; ///< side-1 PV sums and the remaining velocity terms in one kernel
   int MergeL1   = 1; ///< vertex pass + side-0 PV sums inside the level-1 cell kernel
   int Pair      = 1; ///< independent sweeps share a launch
   int FuseL3    = 1; ///< plain RHS: both level-3 kernels in one thread
   int InlineOther = 1; ///< merged level-1 kernel: side-0 PV sums of the cells with one edge fewer inside the sweep
   int TracerPatch = 1; ///< level-3 kernel of the plain RHS: the tracer loop's neighbour values staged through LDS tile patches (CellPVFinalTracerPatchBody)
   int FoldLists = 1; ///< plain RHS: the other valence's final-pass cell list joins the level-3 sweep's launch
   int Alternate  = 0; ///< 1: consecutive dependency levels sweep the mesh in opposite directions (measured: no gain)
   int SendBand   = 1; ///< overlapped RK4 stages: the level-3 kernels skip the halo cells whose results the exchange replaces
   int BandOnComm = 1; ///< overlapped RK4 stages: the band launches run on the communication stream, next to the interior ones
   int ShrinkSweeps = 1; ///< RK4 stages sweep only as many halo layers as the rank still reads (StageUpdate::NCellsL1 ...)
   // ---- mesh tables (read when a HorzMesh is constructed)
   int ForceGeneric = 0; ///< clear every ring-table flag: all kernels in their generic form
   int KeepMaxEdges = 0; ///< keep the mesh file's maxEdges as the table width
   int DomValence   = 1; ///< full sweeps at the valence most cells have
   int NarrowTables = 1; ///< hexagon-dominant meshes with heptagons: second, MaxEdges-1 wide set of cell tables
   // ---- local numbering (read when a Decomp is constructed with a curve order)
   int WaveWindow = 0; ///< >= 16: cells regrouped inside windows of this many consecutive cells so that a wave's 8 cells finish the same edge slots (Decomp.h)
   // ---- measurement probe (KernelCommon.h: SliceWindow; timings only, rim values of the blocks are wrong)
   int ProbeSlice  = 0; ///< 1: the plain fused RHS block by block, per block L1 -> L2 -> L3 (what a cache-blocked walk would read back from the memory-side cache); 2: the same launches level by level (nothing resident): the difference is what the residency is worth
   int ProbeBlocks = 1; ///< blocks of tiles per level chunk under ProbeSlice
   // ---- HIP-graph replay: -1 = as each object's UseGraphs says, 0 = never, 1 = default on
   int Graphs = -1;
   // ---- (appended last: the layout above is what already-built kernel objects index)
   int ValenceSort = 0; ///< 1: k-d local order with the cells of each group's dominant valence first (k-d ordered among themselves), the others after them -- the sweeps' tiles then hold no cell the sweep skips (Decomp.cpp: kdOrder).  Measured: +-0.3 % on the Fibonacci and icosahedral spheres (profiles/r05_ab_valence_sort_*.jsonl): off
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
