// MPAS mesh / initial-state file reader.  The reference reads its mesh through SCORPIO / PIO
// (components/omega/src/base/Decomp.cpp:108-395 readMesh, src/ocn/HorzMesh.cpp:424-523), which is
// outside the hot path and not available here; MPAS-Ocean meshes are NetCDF classic files (CDF-1,
// CDF-2 "64-bit offset", CDF-5 "64-bit data"), whose layout is simple enough to parse directly:
// header (dimensions, attributes, variables with type / shape / file offset) followed by big-endian
// array data.  NcFile reads those three formats; MeshFile maps an MPAS mesh onto GlobalMeshDesc under
// both name conventions the reference accepts ("NCells" / "nCells", "CellsOnCell" / "cellsOnCell", ...;
// Decomp.cpp:136-206, 337-391), converting the file's 1-based indices (0 = none) to 0-based (-1 = none).
#ifndef OMEGA_AMD_MESHIO_H
#define OMEGA_AMD_MESHIO_H

#include "Base.h"
#include "Decomp.h"

#include <map>
#include <string>
#include <vector>

namespace OMEGA {

class NcFile {
 public:
   explicit NcFile(const std::string &Path);
   ~NcFile();
   NcFile(const NcFile &)            = delete;
   NcFile &operator=(const NcFile &) = delete;

   struct Var {
      std::string Name;
      std::vector<int> DimIds;
      int Type = 0;       ///< nc_type: 1 byte, 2 char, 3 short, 4 int, 5 float, 6 double, 7-11 (CDF-5) ubyte..uint64
      I8 VSize = 0;       ///< bytes per record (record variables) or in total, padded
      I8 Begin = 0;       ///< file offset
      bool IsRecord = false;
   };

   int Version = 0; ///< 1, 2 or 5
   I8 NumRecs  = 0;
   bool hasDim(const std::string &Name) const;
   I8 dimLen(const std::string &Name) const; ///< aborts if missing
   bool hasVar(const std::string &Name) const;
   const Var &var(const std::string &Name) const;
   std::vector<I8> shape(const std::string &Name) const; ///< record dimension reported as NumRecs
   /// whole variable (Record < 0) or one record of a record variable, converted to double / int32
   void read(const std::string &Name, std::vector<R8> &Out, I8 Record = -1) const;
   void read(const std::string &Name, std::vector<I4> &Out, I8 Record = -1) const;
   std::vector<std::string> varNames() const;

 private:
   template <class T> void readAs(const std::string &Name, std::vector<T> &Out, I8 Record) const;
   FILE *F = nullptr;
   std::string Path;
   std::vector<std::string> DimNames;
   std::vector<I8> DimLens;
   std::vector<Var> Vars;
   std::map<std::string, int> VarIndex;
   I8 RecSize = 0;
};

/// An MPAS mesh file as the global mesh the decomposition starts from.
class MeshFile {
 public:
   /// opens the file and parses its header; the mesh arrays are read on the first desc() call, so a file that
   /// only holds an initial state / forcing can be opened and read through file().read()
   explicit MeshFile(const std::string &Path);
   const GlobalMeshDesc &desc();
   const NcFile &file() const { return Nc; }

 private:
   NcFile Nc;
   GlobalMeshDesc Desc;
   bool MeshLoaded = false;
   void loadMesh();
   std::map<std::string, std::vector<I4>> IntArrays;
   std::map<std::string, std::vector<R8>> RealArrays;
   const I4 *conn(const std::string &OmegaName, const std::string &MpasName, I8 Expect);
   const R8 *real(const std::string &MpasName, I8 Expect, bool Required);
};

/// Restart file (reference: the "RestartWrite" / "InitialState" IOStreams of Default.yml:91-127 write
/// LayerThickness, NormalVelocity and the tracers through SCORPIO; here: one NetCDF classic CDF-5 file
/// with fixed-size variables layerThickness(nCells, nVertLevels), normalVelocity(nEdges, nVertLevels),
/// tracers(nTracers, nCells, nVertLevels), simulationTime, stepsDone).  The classic layout is fixed by the
/// header, so every rank writes / reads the rows of its own elements at their global positions with
/// plain positioned IO: no gather, and a restarted run may use a different partition.
class RestartFile {
 public:
   /// rank 0: create the file and write the header (and the two scalars)
   static void create(const std::string &Path, I8 NCellsGlobal, I8 NEdgesGlobal, int NVertLevels, int NTracers,
                      R8 SimulationTime, I8 StepsDone);
   explicit RestartFile(const std::string &Path, bool Write);
   ~RestartFile();
   I8 NCells = 0, NEdges = 0;
   int NVertLevels = 0, NTracers = 0;
   R8 SimulationTime = 0;
   I8 StepsDone      = 0;
   /// rows of `Var` ("layerThickness" | "normalVelocity" | "tracers" with Plane = tracer index) for the N
   /// elements with 1-based global ids GlobalID[0..N): Rows is [N][NVertLevels], host memory
   void writeRows(const std::string &Var, int Plane, const I4 *GlobalID, I8 N, const R8 *Rows);
   void readRows(const std::string &Var, int Plane, const I4 *GlobalID, I8 N, R8 *Rows) const;

 private:
   int Fd = -1;
   std::string Path;
   std::map<std::string, I8> Begin;
   I8 rowOffset(const std::string &Var, int Plane, I8 GlobalRow) const;
};

} // namespace OMEGA
#endif
