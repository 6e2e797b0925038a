// See MeshIO.h.  NetCDF classic file format: "The NetCDF Classic Format Specification" (CDF-1 / CDF-2)
// and the PnetCDF CDF-5 extension (64-bit sizes, extra integer types).
#include "MeshIO.h"

#include <cstdio>
#include <cstring>
#include <fcntl.h>
#include <unistd.h>

namespace OMEGA {

namespace {

constexpr int NcDimension = 0x0A, NcVariable = 0x0B, NcAttribute = 0x0C;

struct Reader {
   FILE *F;
   int Version;
   const std::string &Path;
   void bytes(void *Dst, size_t N) {
      if (N && fread(Dst, 1, N, F) != N)
         OMEGA_ABORT("NcFile: unexpected end of header in " + Path);
   }
   unsigned long long be(int N) {
      unsigned char B[8];
      bytes(B, N);
      unsigned long long V = 0;
      for (int I = 0; I < N; ++I)
         V = (V << 8) | B[I];
      return V;
   }
   I8 i4() { return (I8)(int)be(4); }
   I8 nonNeg() { return Version == 5 ? (I8)be(8) : (I8)be(4); }   ///< NON_NEG: INT in CDF-1/2, INT64 in CDF-5
   I8 offset() { return Version == 1 ? (I8)be(4) : (I8)be(8); }   ///< OFFSET: 32 bit in CDF-1
   std::string name() {
      const I8 N = nonNeg();
      if (N < 0 || N > 4096) // NC_MAX_NAME is 256: anything longer is a corrupt header, not a name to allocate
         OMEGA_ABORT("NcFile: implausible name length in the header of " + Path);
      std::string S((size_t)N, '\0');
      bytes(S.data(), (size_t)N);
      skip((4 - N % 4) % 4);
      return S;
   }
   void skip(I8 N) {
      if (N > 0 && fseeko(F, (off_t)N, SEEK_CUR) != 0)
         OMEGA_ABORT("NcFile: seek failed in " + Path);
   }
};

int typeSize(int T) {
   switch (T) {
   case 1: case 2: case 7: return 1;
   case 3: case 8: return 2;
   case 4: case 5: case 9: return 4;
   case 6: case 10: case 11: return 8;
   default: return 0;
   }
}

void skipAttributes(Reader &R) {
   const I8 Tag = R.i4(), N = R.nonNeg();
   if (Tag == 0 && N == 0)
      return;
   if (Tag != NcAttribute)
      OMEGA_ABORT("NcFile: malformed attribute list in " + R.Path);
   for (I8 I = 0; I < N; ++I) {
      R.name();
      const int T   = (int)R.i4();
      const I8 Cnt  = R.nonNeg();
      if (Cnt < 0 || Cnt > ((I8)1 << 40))
         OMEGA_ABORT("NcFile: implausible attribute length in the header of " + R.Path);
      const I8 Size = Cnt * typeSize(T);
      R.skip(Size + (4 - Size % 4) % 4);
   }
}

template <class T> T convert(const unsigned char *P, int Type) {
   auto U = [&](int N) {
      unsigned long long V = 0;
      for (int I = 0; I < N; ++I)
         V = (V << 8) | P[I];
      return V;
   };
   switch (Type) {
   case 1: return (T)(signed char)P[0];
   case 2: case 7: return (T)P[0];
   case 3: return (T)(short)U(2);
   case 8: return (T)(unsigned short)U(2);
   case 4: return (T)(int)U(4);
   case 9: return (T)(unsigned)U(4);
   case 5: {
      const unsigned V = (unsigned)U(4);
      float Fv;
      std::memcpy(&Fv, &V, 4);
      return (T)Fv;
   }
   case 6: {
      const unsigned long long V = U(8);
      double Dv;
      std::memcpy(&Dv, &V, 8);
      return (T)Dv;
   }
   case 10: return (T)(long long)U(8);
   case 11: return (T)U(8);
   default: return T();
   }
}

} // namespace

NcFile::NcFile(const std::string &InPath) : Path(InPath) {
   F = fopen(Path.c_str(), "rb");
   if (!F)
      OMEGA_ABORT("NcFile: cannot open " + Path);
   unsigned char Magic[4];
   if (fread(Magic, 1, 4, F) != 4 || Magic[0] != 'C' || Magic[1] != 'D' || Magic[2] != 'F' ||
       (Magic[3] != 1 && Magic[3] != 2 && Magic[3] != 5)) {
      const bool Hdf = Magic[0] == 0x89 && Magic[1] == 'H';
      fclose(F);
      F = nullptr;
      OMEGA_ABORT("NcFile: " + Path +
                  (Hdf ? " is NetCDF-4/HDF5; convert it with `nccopy -k cdf5` (only the classic formats are read)"
                       : " is not a NetCDF classic file (CDF-1, CDF-2 or CDF-5)"));
   }
   Version = Magic[3];
   Reader R{F, Version, Path};
   NumRecs = R.nonNeg();
   // dim_list
   {
      const I8 Tag = R.i4(), N = R.nonNeg();
      if (!(Tag == 0 && N == 0)) {
         if (Tag != NcDimension)
            OMEGA_ABORT("NcFile: malformed dimension list in " + Path);
         for (I8 I = 0; I < N; ++I) {
            DimNames.push_back(R.name());
            DimLens.push_back(R.nonNeg());
         }
      }
   }
   skipAttributes(R); // global attributes
   // var_list
   {
      const I8 Tag = R.i4(), N = R.nonNeg();
      if (!(Tag == 0 && N == 0)) {
         if (Tag != NcVariable)
            OMEGA_ABORT("NcFile: malformed variable list in " + Path);
         for (I8 I = 0; I < N; ++I) {
            Var V;
            V.Name        = R.name();
            const I8 Rank = R.nonNeg();
            for (I8 D = 0; D < Rank; ++D) {
               const I8 Id = R.nonNeg();
               if (Id < 0 || Id >= (I8)DimLens.size())
                  OMEGA_ABORT("NcFile: bad dimension id in variable " + V.Name);
               V.DimIds.push_back((int)Id);
            }
            skipAttributes(R);
            V.Type     = (int)R.i4();
            V.VSize    = R.nonNeg();
            V.Begin    = R.offset();
            V.IsRecord = Rank > 0 && DimLens[V.DimIds[0]] == 0;
            if (typeSize(V.Type) == 0)
               OMEGA_ABORT("NcFile: unsupported type in variable " + V.Name);
            VarIndex[V.Name] = (int)Vars.size();
            Vars.push_back(V);
         }
      }
   }
   // record size: sum of the (padded) per-record sizes; a single record variable is not padded
   int NRec = 0;
   for (const Var &V : Vars)
      if (V.IsRecord) {
         RecSize += V.VSize;
         ++NRec;
      }
   if (NRec == 1)
      for (const Var &V : Vars)
         if (V.IsRecord) {
            I8 N = typeSize(V.Type);
            for (size_t D = 1; D < V.DimIds.size(); ++D)
               N *= DimLens[V.DimIds[D]];
            RecSize = N;
         }
}

NcFile::~NcFile() {
   if (F)
      fclose(F);
}

bool NcFile::hasDim(const std::string &Name) const {
   for (const auto &D : DimNames)
      if (D == Name)
         return true;
   return false;
}
I8 NcFile::dimLen(const std::string &Name) const {
   for (size_t I = 0; I < DimNames.size(); ++I)
      if (DimNames[I] == Name)
         return DimLens[I] == 0 ? NumRecs : DimLens[I];
   OMEGA_ABORT("NcFile: no dimension " + Name + " in " + Path);
}
bool NcFile::hasVar(const std::string &Name) const { return VarIndex.count(Name) != 0; }
const NcFile::Var &NcFile::var(const std::string &Name) const {
   auto It = VarIndex.find(Name);
   if (It == VarIndex.end())
      OMEGA_ABORT("NcFile: no variable " + Name + " in " + Path);
   return Vars[It->second];
}
std::vector<I8> NcFile::shape(const std::string &Name) const {
   const Var &V = var(Name);
   std::vector<I8> S;
   for (int D : V.DimIds)
      S.push_back(DimLens[D] == 0 ? NumRecs : DimLens[D]);
   return S;
}
std::vector<std::string> NcFile::varNames() const {
   std::vector<std::string> N;
   for (const Var &V : Vars)
      N.push_back(V.Name);
   return N;
}

template <class T> void NcFile::readAs(const std::string &Name, std::vector<T> &Out, I8 Record) const {
   const Var &V = var(Name);
   I8 PerRec    = 1; // elements per record (record variables) or in total
   for (size_t D = V.IsRecord ? 1 : 0; D < V.DimIds.size(); ++D)
      PerRec *= DimLens[V.DimIds[D]];
   const int Ts = typeSize(V.Type);
   I8 NRec = 1, First = 0;
   if (V.IsRecord) {
      if (Record >= NumRecs)
         OMEGA_ABORT("NcFile: record out of range for " + Name);
      NRec  = Record < 0 ? NumRecs : 1;
      First = Record < 0 ? 0 : Record;
   }
   Out.resize((size_t)(PerRec * NRec));
   std::vector<unsigned char> Buf((size_t)(PerRec * Ts));
   for (I8 Rr = 0; Rr < NRec; ++Rr) {
      const I8 Off = V.Begin + (V.IsRecord ? (First + Rr) * RecSize : 0);
      if (fseeko(F, (off_t)Off, SEEK_SET) != 0 || (Buf.size() && fread(Buf.data(), 1, Buf.size(), F) != Buf.size()))
         OMEGA_ABORT("NcFile: short read of " + Name + " in " + Path);
      for (I8 I = 0; I < PerRec; ++I)
         Out[(size_t)(Rr * PerRec + I)] = convert<T>(&Buf[(size_t)I * Ts], V.Type);
   }
}
void NcFile::read(const std::string &Name, std::vector<R8> &Out, I8 Record) const { readAs<R8>(Name, Out, Record); }
void NcFile::read(const std::string &Name, std::vector<I4> &Out, I8 Record) const { readAs<I4>(Name, Out, Record); }

// ---------------------------------------------------------------------------------------
static I8 dimEither(const NcFile &Nc, const std::string &Omega, const std::string &Mpas) {
   // Decomp.cpp:136-206: the Omega name first, then the older MPAS name
   if (Nc.hasDim(Omega))
      return Nc.dimLen(Omega);
   if (Nc.hasDim(Mpas))
      return Nc.dimLen(Mpas);
   OMEGA_ABORT("MeshFile: neither dimension " + Omega + " nor " + Mpas + " found");
}

const I4 *MeshFile::conn(const std::string &Omega, const std::string &Mpas, I8 Expect) {
   const std::string Name = Nc.hasVar(Omega) ? Omega : Mpas; // Decomp.cpp:337-391
   std::vector<I4> &A     = IntArrays[Mpas];
   Nc.read(Name, A);
   if ((I8)A.size() != Expect)
      OMEGA_ABORT("MeshFile: " + Name + " has " + std::to_string(A.size()) + " entries, expected " +
                  std::to_string(Expect));
   for (I4 &V : A)
      V -= 1; // 1-based with 0 = none  ->  0-based with -1 = none (Decomp.cpp:553-574)
   return A.data();
}
const R8 *MeshFile::real(const std::string &Mpas, I8 Expect, bool Required) {
   if (!Nc.hasVar(Mpas)) {
      if (Required)
         OMEGA_ABORT("MeshFile: variable " + Mpas + " not found");
      std::vector<R8> &Z = RealArrays[Mpas];
      Z.assign((size_t)Expect, 0.0);
      return Z.data();
   }
   std::vector<R8> &A = RealArrays[Mpas];
   Nc.read(Mpas, A);
   if ((I8)A.size() != Expect)
      OMEGA_ABORT("MeshFile: " + Mpas + " has " + std::to_string(A.size()) + " entries, expected " +
                  std::to_string(Expect));
   return A.data();
}

MeshFile::MeshFile(const std::string &Path) : Nc(Path) {}

const GlobalMeshDesc &MeshFile::desc() {
   if (!MeshLoaded) {
      loadMesh();
      MeshLoaded = true;
   }
   return Desc;
}

void MeshFile::loadMesh() {
   GlobalMeshDesc &D = Desc;
   D.NCells       = (I4)dimEither(Nc, "NCells", "nCells");
   D.NEdges       = (I4)dimEither(Nc, "NEdges", "nEdges");
   D.NVertices    = (I4)dimEither(Nc, "NVertices", "nVertices");
   D.MaxEdges     = (I4)dimEither(Nc, "MaxEdges", "maxEdges");
   D.VertexDegree = (I4)dimEither(Nc, "VertexDegree", "vertexDegree");
   const I8 NC = D.NCells, NE = D.NEdges, NV = D.NVertices, ME = D.MaxEdges, VD = D.VertexDegree;
   D.CellsOnCell    = conn("CellsOnCell", "cellsOnCell", NC * ME);
   D.EdgesOnCell    = conn("EdgesOnCell", "edgesOnCell", NC * ME);
   D.VerticesOnCell = conn("VerticesOnCell", "verticesOnCell", NC * ME);
   D.CellsOnEdge    = conn("CellsOnEdge", "cellsOnEdge", NE * 2);
   D.VerticesOnEdge = conn("VerticesOnEdge", "verticesOnEdge", NE * 2);
   D.EdgesOnEdge    = conn("EdgesOnEdge", "edgesOnEdge", NE * 2 * ME);
   D.CellsOnVertex  = conn("CellsOnVertex", "cellsOnVertex", NV * VD);
   D.EdgesOnVertex  = conn("EdgesOnVertex", "edgesOnVertex", NV * VD);
   // MPAS pads edgesOnCell & co. beyond nEdgesOnCell with the last valid (or any) index: blank them
   if (Nc.hasVar("nEdgesOnCell") || Nc.hasVar("NEdgesOnCell")) {
      std::vector<I4> N;
      Nc.read(Nc.hasVar("NEdgesOnCell") ? "NEdgesOnCell" : "nEdgesOnCell", N);
      OMEGA_REQUIRE((I8)N.size() == NC, "MeshFile: nEdgesOnCell has the wrong length");
      for (I8 C = 0; C < NC; ++C)
         if (N[(size_t)C] < 0 || N[(size_t)C] > ME)
            OMEGA_ABORT("MeshFile: nEdgesOnCell(" + std::to_string(C) + ") = " + std::to_string(N[(size_t)C]) +
                        " is outside [0, maxEdges]");
      for (const char *Nm : {"cellsOnCell", "edgesOnCell", "verticesOnCell"}) {
         std::vector<I4> &A = IntArrays[Nm];
         for (I8 C = 0; C < NC; ++C)
            for (I8 J = N[(size_t)C]; J < ME; ++J)
               A[(size_t)(C * ME + J)] = -1;
      }
   }
   if (Nc.hasVar("nEdgesOnEdge") || Nc.hasVar("NEdgesOnEdge")) {
      std::vector<I4> Ne;
      Nc.read(Nc.hasVar("NEdgesOnEdge") ? "NEdgesOnEdge" : "nEdgesOnEdge", Ne);
      OMEGA_REQUIRE((I8)Ne.size() == NE, "MeshFile: nEdgesOnEdge has the wrong length");
      std::vector<I4> &A = IntArrays["edgesOnEdge"];
      for (I8 E = 0; E < NE; ++E) {
         if (Ne[(size_t)E] < 0 || Ne[(size_t)E] > 2 * ME)
            OMEGA_ABORT("MeshFile: nEdgesOnEdge(" + std::to_string(E) + ") = " + std::to_string(Ne[(size_t)E]) +
                        " is outside [0, 2*maxEdges]");
         for (I8 J = Ne[(size_t)E]; J < 2 * ME; ++J)
            A[(size_t)(E * 2 * ME + J)] = -1;
      }
   }
   // geometry (HorzMesh.cpp:424-523)
   D.XCell = real("xCell", NC, true), D.YCell = real("yCell", NC, true), D.ZCell = real("zCell", NC, true);
   D.LonCell = real("lonCell", NC, false), D.LatCell = real("latCell", NC, false);
   D.XEdge = real("xEdge", NE, true), D.YEdge = real("yEdge", NE, true), D.ZEdge = real("zEdge", NE, true);
   D.LonEdge = real("lonEdge", NE, false), D.LatEdge = real("latEdge", NE, false);
   D.XVertex = real("xVertex", NV, true), D.YVertex = real("yVertex", NV, true), D.ZVertex = real("zVertex", NV, true);
   D.LonVertex = real("lonVertex", NV, false), D.LatVertex = real("latVertex", NV, false);
   D.AreaCell = real("areaCell", NC, true), D.AreaTriangle = real("areaTriangle", NV, true);
   D.KiteAreasOnVertex = real("kiteAreasOnVertex", NV * VD, true);
   D.DcEdge = real("dcEdge", NE, true), D.DvEdge = real("dvEdge", NE, true), D.AngleEdge = real("angleEdge", NE, true);
   D.WeightsOnEdge = real("weightsOnEdge", NE * 2 * ME, true);
   D.FCell = real("fCell", NC, false), D.FEdge = real("fEdge", NE, false), D.FVertex = real("fVertex", NV, false);
   D.BottomDepth = real("bottomDepth", NC, false);
}

// ---------------------------------------------------------------------------------------
// RestartFile
namespace {
void putBE(std::vector<unsigned char> &B, unsigned long long V, int N) {
   for (int I = N - 1; I >= 0; --I)
      B.push_back((unsigned char)((V >> (8 * I)) & 0xff));
}
void putName(std::vector<unsigned char> &B, const std::string &S) {
   putBE(B, S.size(), 8);
   B.insert(B.end(), S.begin(), S.end());
   for (size_t I = S.size(); I % 4; ++I)
      B.push_back(0);
}
void swapCopy(unsigned char *Dst, const R8 *Src, I8 N) { // host doubles -> big-endian bytes (and back)
   for (I8 I = 0; I < N; ++I) {
      unsigned char T[8];
      std::memcpy(T, &Src[I], 8);
      for (int J = 0; J < 8; ++J)
         Dst[I * 8 + J] = T[7 - J];
   }
}
} // namespace

void RestartFile::create(const std::string &Path, I8 NC, I8 NE, int K, int NT, R8 SimTime, I8 Steps) {
   struct V {
      std::string Name;
      std::vector<int> Dims;
      int Type;
      I8 Bytes;
   };
   const std::vector<std::pair<std::string, I8>> Dims{{"nCells", NC}, {"nEdges", NE}, {"nVertLevels", K},
                                                      {"nTracers", NT > 0 ? NT : 1}};
   const std::vector<V> Vars{{"simulationTime", {}, 6, 8},
                             {"stepsDone", {}, 10, 8},
                             {"layerThickness", {0, 2}, 6, NC * K * 8},
                             {"normalVelocity", {1, 2}, 6, NE * K * 8},
                             {"tracers", {3, 0, 2}, 6, (I8)(NT > 0 ? NT : 1) * NC * K * 8}};
   auto Header = [&](const std::vector<I8> &Begins) {
      std::vector<unsigned char> B{'C', 'D', 'F', 5};
      putBE(B, 0, 8); // numrecs
      putBE(B, 0x0A, 4), putBE(B, Dims.size(), 8);
      for (auto &D : Dims)
         putName(B, D.first), putBE(B, (unsigned long long)D.second, 8);
      putBE(B, 0, 4), putBE(B, 0, 8); // no global attributes
      putBE(B, 0x0B, 4), putBE(B, Vars.size(), 8);
      for (size_t I = 0; I < Vars.size(); ++I) {
         putName(B, Vars[I].Name);
         putBE(B, Vars[I].Dims.size(), 8);
         for (int D : Vars[I].Dims)
            putBE(B, (unsigned long long)D, 8);
         putBE(B, 0, 4), putBE(B, 0, 8); // no attributes
         putBE(B, (unsigned long long)Vars[I].Type, 4);
         putBE(B, (unsigned long long)Vars[I].Bytes, 8);
         putBE(B, (unsigned long long)Begins[I], 8);
      }
      return B;
   };
   std::vector<I8> Begins(Vars.size(), 0);
   const I8 HLen = (I8)Header(Begins).size();
   I8 Off        = HLen;
   for (size_t I = 0; I < Vars.size(); ++I) {
      Begins[I] = Off;
      Off += Vars[I].Bytes;
   }
   const std::vector<unsigned char> H = Header(Begins);
   FILE *F = fopen(Path.c_str(), "wb");
   if (!F)
      OMEGA_ABORT("RestartFile: cannot create " + Path);
   bool Ok = fwrite(H.data(), 1, H.size(), F) == H.size();
   unsigned char T[16];
   swapCopy(T, &SimTime, 1);
   for (int J = 0; J < 8; ++J)
      T[8 + J] = (unsigned char)(((unsigned long long)Steps >> (8 * (7 - J))) & 0xff);
   Ok = Ok && fwrite(T, 1, 16, F) == 16;
   // extend to the full size so that every rank can write its rows at their final positions
   Ok = Ok && fseeko(F, (off_t)(Off - 1), SEEK_SET) == 0 && fputc(0, F) != EOF;
   Ok = (fclose(F) == 0) && Ok;
   if (!Ok)
      OMEGA_ABORT("RestartFile: error writing the header of " + Path);
}

RestartFile::RestartFile(const std::string &InPath, bool Write) : Path(InPath) {
   {
      NcFile Nc(Path); // header through the reader
      NCells = Nc.dimLen("nCells"), NEdges = Nc.dimLen("nEdges");
      NVertLevels = (int)Nc.dimLen("nVertLevels"), NTracers = (int)Nc.dimLen("nTracers");
      for (const char *N : {"layerThickness", "normalVelocity", "tracers"})
         Begin[N] = Nc.var(N).Begin;
      std::vector<R8> T;
      Nc.read("simulationTime", T);
      SimulationTime = T[0];
      Nc.read("stepsDone", T);
      StepsDone = (I8)T[0];
   }
   Fd = open(Path.c_str(), Write ? O_RDWR : O_RDONLY);
   if (Fd < 0)
      OMEGA_ABORT("RestartFile: cannot open " + Path);
}
RestartFile::~RestartFile() {
   if (Fd >= 0)
      close(Fd);
}
I8 RestartFile::rowOffset(const std::string &Var, int Plane, I8 Row) const {
   auto It = Begin.find(Var);
   if (It == Begin.end())
      OMEGA_ABORT("RestartFile: no variable " + Var);
   const I8 Rows = Var == "normalVelocity" ? NEdges : NCells;
   if (Row < 0 || Row >= Rows || Plane < 0 || (Var == "tracers" ? Plane >= NTracers : Plane != 0))
      OMEGA_ABORT("RestartFile: row / plane out of range for " + Var);
   return It->second + ((I8)Plane * Rows + Row) * NVertLevels * 8;
}
void RestartFile::writeRows(const std::string &Var, int Plane, const I4 *GlobalID, I8 N, const R8 *Rows) {
   std::vector<unsigned char> Buf((size_t)NVertLevels * 8);
   for (I8 I = 0; I < N; ++I) {
      swapCopy(Buf.data(), Rows + I * NVertLevels, NVertLevels);
      if (pwrite(Fd, Buf.data(), Buf.size(), (off_t)rowOffset(Var, Plane, GlobalID[I] - 1)) != (ssize_t)Buf.size())
         OMEGA_ABORT("RestartFile: short write to " + Path);
   }
}
void RestartFile::readRows(const std::string &Var, int Plane, const I4 *GlobalID, I8 N, R8 *Rows) const {
   std::vector<unsigned char> Buf((size_t)NVertLevels * 8);
   for (I8 I = 0; I < N; ++I) {
      if (pread(Fd, Buf.data(), Buf.size(), (off_t)rowOffset(Var, Plane, GlobalID[I] - 1)) != (ssize_t)Buf.size())
         OMEGA_ABORT("RestartFile: short read from " + Path);
      for (int K = 0; K < NVertLevels; ++K) {
         unsigned char T[8];
         for (int J = 0; J < 8; ++J)
            T[J] = Buf[(size_t)K * 8 + 7 - J];
         std::memcpy(&Rows[I * NVertLevels + K], T, 8);
      }
   }
}

} // namespace OMEGA
