// History.h -- history output of the state, the tracers and the auxiliary fields.
//
// Reference: the "History" IOStream (components/omega/configs/Default.yml:115-127) writes the fields and field
// groups named in its `Contents` (State, Tracers, SshCell, ...) through IOStream / SCORPIO; the auxiliary fields
// are registered under their array names with long names and units in
// components/omega/src/ocn/auxiliaryVars/{Kinetic,LayerThickness,Vorticity,VelocityDel2,Tracer,WindForcing}AuxVars.cpp
// (registerFields) and collected in the group "AuxiliaryState" (AuxiliaryState.cpp:36-47); the state fields are
// "NormalVelocity" and "LayerThickness" in the group "State" (OceanState.cpp:190-234).
//
// Here: one NetCDF classic CDF-5 file per dump with the reference's dimension names (NCells, NEdges, NVertices,
// NVertLayers, NTracers) and field names, long_name / units attributes from the reference, every rank writing the
// rows of its OWNED elements at their global positions (plain positioned IO, as RestartFile).  The fused RHS does
// not materialise the edge-located auxiliary arrays and SshCell: writeHistory therefore
// recomputes the whole AuxiliaryState from the state being written (AuxiliaryState::computeAll) before it copies
// anything, so a dump never holds stale fields.
#ifndef OMEGA_AMD_HISTORY_H
#define OMEGA_AMD_HISTORY_H

#include "AuxiliaryState.h"
#include "Decomp.h"
#include "OceanState.h"

namespace OMEGA {

struct HistoryField {
   std::string Name, LongName, Units;
   MeshElement Elem;
   bool HasLevels; ///< (NX, NVertLayers) or (NX)
   bool PerTracer; ///< leading NTracers dimension
};
/// every field a Contents entry can name, with the reference's metadata
const std::vector<HistoryField> &historyCatalogue();
/// expands group names ("State", "Tracers", "AuxiliaryState") and checks field names
std::vector<HistoryField> expandHistoryContents(const std::string &ContentsCsv);

/// One dump.  CreateFile: this rank writes the header first (rank 0; the caller synchronises the ranks between
/// the creation and the other ranks' calls).  Recomputes the auxiliary state from (State, Tracers) at time level
/// `TimeLevel` on stream S and synchronises it.  Returns the number of variables written.
int writeHistory(const std::string &Path, const Decomp *D, const OceanState *State, const TracerStore *Trc,
                 AuxiliaryState *Aux, const std::string &ContentsCsv, R8 SimTimeSeconds, int TimeLevel, bool CreateFile,
                 hipStream_t S);

} // namespace OMEGA
#endif
