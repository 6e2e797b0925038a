// Partition.h -- built-in partitioners for the cell graph (CellsOnCell).
//
// The reference calls METIS_PartGraphKway on the unweighted cell adjacency graph
// (components/omega/src/base/Decomp.cpp:868-1000).  METIS is not available here; two built-in methods:
//   * RCB   -- recursive coordinate bisection over the cell centres (Decomp::partitionRCB): balanced and compact on
//              quasi-uniform meshes, blind to the graph;
//   * Graph -- recursive GRAPH bisection (breadth-first level structure from a pseudo-peripheral cell, split at the
//              median, Fiduccia-Mattheyses boundary refinement of every bisection) followed by a greedy k-way
//              boundary refinement; needs no coordinates, minimises the edge cut (= halo size) under a 3 % imbalance
//              tolerance (METIS' default ufactor), and follows variable resolution because it only sees adjacency.
// A caller-supplied cell -> task vector (METIS graph.info.part.N files) is still honoured by Decomp as is.
#ifndef OMEGA_AMD_PARTITION_H
#define OMEGA_AMD_PARTITION_H

#include "Decomp.h"

namespace OMEGA {

/// recursive coordinate bisection of the cell centres (ties broken by global id)
void partitionRCB(const GlobalMeshDesc &G, I4 NParts, std::vector<I4> &CellTask);
/// CellTask[c] in [0, NParts) for every global cell.  Deterministic.
void partitionGraph(const GlobalMeshDesc &G, I4 NParts, std::vector<I4> &CellTask);
/// number of cell-graph edges whose two cells lie in different parts
I8 edgeCut(const GlobalMeshDesc &G, const std::vector<I4> &CellTask);

} // namespace OMEGA
#endif
