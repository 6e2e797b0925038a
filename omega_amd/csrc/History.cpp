// History.cpp -- see History.h.
#include "History.h"
#include "MeshIO.h"

#include <cstring>
#include <fcntl.h>
#include <sstream>
#include <sys/mman.h>
#include <unistd.h>

namespace OMEGA {

const std::vector<HistoryField> &historyCatalogue() {
   // names = the reference's array labels; long names / units from auxiliaryVars/*.cpp registerFields and
   // OceanState.cpp:190-215
   static const std::vector<HistoryField> C{
       {"LayerThickness", "Thickness of layer on cell center", "m", OnCell, true, false},
       {"NormalVelocity", "Velocity component normal to edge", "m/s", OnEdge, true, false},
       {"Tracers", "tracer concentrations", "", OnCell, true, true},
       {"KineticEnergyCell", "kinetic energy of horizontal velocity on cells", "m^2 s^-2", OnCell, true, false},
       {"VelocityDivCell", "divergence of horizontal velocity", "s^-1", OnCell, true, false},
       {"FluxLayerThickEdge", "layer thickness used for fluxes through edges. May be centered, upwinded, or a combination of the two.", "m", OnEdge, true, false},
       {"MeanLayerThickEdge", "layer thickness averaged from cell center to edges", "m", OnEdge, true, false},
       {"SshCell", "sea surface height at cell center", "m", OnCell, true, false},
       {"RelVortVertex", "curl of horizontal velocity, defined at vertices", "s^-1", OnVertex, true, false},
       {"NormRelVortVertex", "curl of horizontal velocity divided by layer thickness", "m^-1 s^-1", OnVertex, true, false},
       {"NormPlanetVortVertex", "earth's rotational rate (Coriolis parameter, f) divided by layer thickness", "m^-1 s^-1", OnVertex, true, false},
       {"NormRelVortEdge", "curl of horizontal velocity divided by layer thickness, averaged from vertices to edges", "m^-1 s^-1", OnEdge, true, false},
       {"NormPlanetVortEdge", "earth's rotational rate (Coriolis parameter, f) divided by layer thickness, averaged from vertices to edges", "m^-1 s^-1", OnEdge, true, false},
       {"Del2Edge", "laplacian of horizontal velocity on edges", "m^-1 s^-1", OnEdge, true, false},
       {"Del2DivCell", "divergence of laplacian of horizontal velocity on cells", "m^-2 s^-1", OnCell, true, false},
       {"Del2RelVortVertex", "laplacian of relative vorticity at vertices", "m^-2 s^-1", OnVertex, true, false},
       {"HTracersEdge", "thickness-weighted tracers at edges. May be centered, upwinded, or a combination of the two.", "", OnEdge, true, true},
       {"Del2TracersCell", "laplacian of thickness-weighted tracers at cell center", "", OnCell, true, true},
       {"NormalStressEdge", "wind stress component normal to edge", "N m^{-2}", OnEdge, false, false},
       {"ZonalStressCell", "zonal wind stress", "N m^{-2}", OnCell, false, false},
       {"MeridStressCell", "meridional wind stress", "N m^{-2}", OnCell, false, false}};
   return C;
}

std::vector<HistoryField> expandHistoryContents(const std::string &Csv) {
   const auto &Cat = historyCatalogue();
   std::vector<std::string> Names;
   std::stringstream SS(Csv);
   std::string Tok;
   while (std::getline(SS, Tok, ',')) {
      const size_t A = Tok.find_first_not_of(" \t"), B = Tok.find_last_not_of(" \t");
      if (A == std::string::npos)
         continue;
      Tok = Tok.substr(A, B - A + 1);
      if (Tok == "State") // OceanState.cpp:224-234
         Names.push_back("LayerThickness"), Names.push_back("NormalVelocity");
      else if (Tok == "AuxiliaryState" || Tok == "AuxState") // AuxiliaryState.cpp:36-47
         for (size_t I = 3; I < Cat.size(); ++I)
            Names.push_back(Cat[I].Name);
      else
         Names.push_back(Tok);
   }
   std::vector<HistoryField> Out;
   for (const std::string &N : Names) {
      bool Found = false, Dup = false;
      for (const HistoryField &O : Out)
         Dup |= O.Name == N;
      for (const HistoryField &F : Cat)
         if (F.Name == N && !Dup) {
            Out.push_back(F);
            Found = true;
         }
      if (!Found && !Dup)
         OMEGA_ABORT("History: no field or field group named " + N);
   }
   OMEGA_REQUIRE(!Out.empty(), "History: empty Contents");
   return Out;
}

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
void putTextAttr(std::vector<unsigned char> &B, const std::string &Name, const std::string &Val) {
   putName(B, Name);
   putBE(B, 2, 4); // NC_CHAR
   putBE(B, Val.size(), 8);
   B.insert(B.end(), Val.begin(), Val.end());
   for (size_t I = Val.size(); I % 4; ++I)
      B.push_back(0);
}
struct VarLayout {
   I8 Begin, Rows; ///< rows per plane
   int Planes, RowLen;
};
// dims: 0 NCells 1 NEdges 2 NVertices 3 NVertLayers 4 NTracers
std::vector<unsigned char> header(const std::vector<HistoryField> &F, const I8 N[3], int K, int NT,
                                  const std::vector<I8> &Begins, std::vector<I8> *Bytes) {
   std::vector<unsigned char> B{'C', 'D', 'F', 5};
   putBE(B, 0, 8);
   const char *DimNames[5] = {"NCells", "NEdges", "NVertices", "NVertLayers", "NTracers"};
   const I8 DimLen[5]      = {N[0], N[1], N[2], K, NT > 0 ? NT : 1};
   putBE(B, 0x0A, 4), putBE(B, 5, 8);
   for (int I = 0; I < 5; ++I)
      putName(B, DimNames[I]), putBE(B, (unsigned long long)DimLen[I], 8);
   putBE(B, 0, 4), putBE(B, 0, 8); // no global attributes
   putBE(B, 0x0B, 4), putBE(B, F.size() + 1, 8);
   putName(B, "SimulationTime"), putBE(B, 0, 8);
   putBE(B, 0x0C, 4), putBE(B, 1, 8), putTextAttr(B, "units", "seconds since the reference time");
   putBE(B, 6, 4), putBE(B, 8, 8), putBE(B, (unsigned long long)Begins[0], 8);
   if (Bytes)
      Bytes->assign(1, 8);
   for (size_t I = 0; I < F.size(); ++I) {
      putName(B, F[I].Name);
      std::vector<int> D;
      if (F[I].PerTracer)
         D.push_back(4);
      D.push_back((int)F[I].Elem);
      if (F[I].HasLevels)
         D.push_back(3);
      putBE(B, D.size(), 8);
      for (int X : D)
         putBE(B, (unsigned long long)X, 8);
      putBE(B, 0x0C, 4), putBE(B, 2, 8);
      putTextAttr(B, "long_name", F[I].LongName), putTextAttr(B, "units", F[I].Units);
      const I8 Sz = (F[I].PerTracer ? DimLen[4] : 1) * N[F[I].Elem] * (F[I].HasLevels ? K : 1) * 8;
      putBE(B, 6, 4), putBE(B, (unsigned long long)Sz, 8), putBE(B, (unsigned long long)Begins[I + 1], 8);
      if (Bytes)
         Bytes->push_back(Sz);
   }
   return B;
}
} // namespace

int writeHistory(const std::string &Path, const Decomp *D, const OceanState *State, const TracerStore *Trc,
                 AuxiliaryState *Aux, const std::string &Csv, R8 SimTime, int TimeLevel, bool CreateFile, hipStream_t S) {
   OMEGA_REQUIRE(D && State && Aux, "writeHistory: missing object");
   const std::vector<HistoryField> F = expandHistoryContents(Csv);
   const int K = State->NVertLayers, NT = Trc ? Trc->NTracers : 0;
   OMEGA_REQUIRE(Aux->NTracers == NT && Aux->NVertLayers == K,
                 "writeHistory: the auxiliary state was created for another tracer / level count");
   const I8 NG[3] = {D->NCellsGlobal, D->NEdgesGlobal, D->NVerticesGlobal};
   // layout (every rank derives the same one)
   std::vector<I8> Begins(F.size() + 1, 0), Bytes;
   const I8 HLen = (I8)header(F, NG, K, NT, Begins, &Bytes).size();
   I8 Off        = HLen;
   for (size_t I = 0; I < Begins.size(); ++I) {
      Begins[I] = Off;
      Off += Bytes[I];
   }
   if (CreateFile) {
      const std::vector<unsigned char> H = header(F, NG, K, NT, Begins, nullptr);
      FILE *Fp = fopen(Path.c_str(), "wb");
      if (!Fp)
         OMEGA_ABORT("History: cannot create " + Path);
      bool Ok = fwrite(H.data(), 1, H.size(), Fp) == H.size();
      unsigned char T[8], Tm[8];
      std::memcpy(Tm, &SimTime, 8);
      for (int J = 0; J < 8; ++J)
         T[J] = Tm[7 - J];
      Ok = Ok && fwrite(T, 1, 8, Fp) == 8;
      Ok = Ok && fseeko(Fp, (off_t)(Off - 1), SEEK_SET) == 0 && fputc(0, Fp) != EOF;
      Ok = (fclose(Fp) == 0) && Ok;
      if (!Ok)
         OMEGA_ABORT("History: error writing the header of " + Path);
   }
   // every auxiliary field from THIS state (the fused RHS leaves most of them un-materialised)
   Array3DReal TrArr;
   if (Trc && NT > 0)
      OMEGA_REQUIRE(Trc->getAll(TrArr, TimeLevel) == 0, "writeHistory: bad tracer time level");
   else
      TrArr = Aux->TracerAux.Del2TracersCell; // NT == 0: never read
   Aux->computeAll(State, TrArr, TimeLevel, TimeLevel, S);
   HIP_CHECK(hipStreamSynchronize(S));

   Array2DReal H, U;
   OMEGA_REQUIRE(State->getLayerThickness(H, TimeLevel) == 0 && State->getNormalVelocity(U, TimeLevel) == 0,
                 "writeHistory: bad time level");
   const int Fd = open(Path.c_str(), O_RDWR);
   if (Fd < 0)
      OMEGA_ABORT("History: cannot open " + Path);
   // rows land at scattered global positions (millions of 8*K-byte pieces for a QU30-sized dump): write them through
   // a shared mapping of the file instead of one pwrite each
   unsigned char *Map = static_cast<unsigned char *>(mmap(nullptr, (size_t)Off, PROT_READ | PROT_WRITE, MAP_SHARED, Fd, 0));
   if (Map == MAP_FAILED) {
      close(Fd);
      OMEGA_ABORT("History: cannot map " + Path);
   }
   const I4 NOwned[3]        = {D->NCellsOwned, D->NEdgesOwned, D->NVerticesOwned};
   const HostArrayI4 *IDs[3] = {&D->CellIDH, &D->EdgeIDH, &D->VertexIDH};
   int NWritten              = 0;
   try {
      for (size_t I = 0; I < F.size(); ++I) {
         const HistoryField &Fd_ = F[I];
         // locate the device array: (pointer, planes, rows per plane, row length, pitch)
         Real *Ptr = nullptr;
         int Planes = 1, RowsSize = 0, RowLen = Fd_.HasLevels ? K : 1, Pitch = 0;
         auto Use2 = [&](const Array2DReal &A) { Ptr = A.Ptr, RowsSize = A.Ext[0], Pitch = A.Pitch; };
         auto Use3 = [&](const Array3DReal &A) { Ptr = A.Ptr, Planes = NT, RowsSize = A.Ext[1], Pitch = A.Pitch; };
         auto Use1 = [&](const Array1DReal &A) { Ptr = A.Ptr, RowsSize = A.Ext[0], Pitch = 1; };
         const std::string &Nm = Fd_.Name;
         if (Nm == "LayerThickness") Use2(H);
         else if (Nm == "NormalVelocity") Use2(U);
         else if (Nm == "Tracers") Use3(TrArr);
         else if (Nm == "KineticEnergyCell") Use2(Aux->KineticAux.KineticEnergyCell);
         else if (Nm == "VelocityDivCell") Use2(Aux->KineticAux.VelocityDivCell);
         else if (Nm == "FluxLayerThickEdge") Use2(Aux->LayerThicknessAux.FluxLayerThickEdge);
         else if (Nm == "MeanLayerThickEdge") Use2(Aux->LayerThicknessAux.MeanLayerThickEdge);
         else if (Nm == "SshCell") Use2(Aux->LayerThicknessAux.SshCell);
         else if (Nm == "RelVortVertex") Use2(Aux->VorticityAux.RelVortVertex);
         else if (Nm == "NormRelVortVertex") Use2(Aux->VorticityAux.NormRelVortVertex);
         else if (Nm == "NormPlanetVortVertex") Use2(Aux->VorticityAux.NormPlanetVortVertex);
         else if (Nm == "NormRelVortEdge") Use2(Aux->VorticityAux.NormRelVortEdge);
         else if (Nm == "NormPlanetVortEdge") Use2(Aux->VorticityAux.NormPlanetVortEdge);
         else if (Nm == "Del2Edge") Use2(Aux->VelocityDel2Aux.Del2Edge);
         else if (Nm == "Del2DivCell") Use2(Aux->VelocityDel2Aux.Del2DivCell);
         else if (Nm == "Del2RelVortVertex") Use2(Aux->VelocityDel2Aux.Del2RelVortVertex);
         else if (Nm == "HTracersEdge") Use3(Aux->TracerAux.HTracersEdge);
         else if (Nm == "Del2TracersCell") Use3(Aux->TracerAux.Del2TracersCell);
         else if (Nm == "NormalStressEdge") Use1(Aux->WindForcingAux.NormalStressEdge);
         else if (Nm == "ZonalStressCell") Use1(Aux->WindForcingAux.ZonalStressCell);
         else if (Nm == "MeridStressCell") Use1(Aux->WindForcingAux.MeridStressCell);
         if (Fd_.PerTracer && NT == 0)
            continue; // nothing to write (the variable stays zero-filled)
         const int E  = (int)Fd_.Elem;
         const I4 No  = NOwned[E];
         std::vector<R8> Host((size_t)No * RowLen);
         for (int P = 0; P < Planes; ++P) {
            copyRowsToHost(Host.data(), Ptr + (size_t)P * RowsSize * Pitch, Pitch, (size_t)No, RowLen);
            for (I4 R = 0; R < No; ++R) {
               const I8 Pos = Begins[I + 1] + (((I8)P * NG[E]) + ((*IDs[E])(R) - 1)) * RowLen * 8;
               unsigned char *Dst = Map + Pos;
               for (int L = 0; L < RowLen; ++L) { // big-endian doubles
                  unsigned char T[8];
                  std::memcpy(T, &Host[(size_t)R * RowLen + L], 8);
                  for (int J = 0; J < 8; ++J)
                     Dst[(size_t)L * 8 + J] = T[7 - J];
               }
            }
         }
         ++NWritten;
      }
   } catch (...) {
      munmap(Map, (size_t)Off);
      close(Fd);
      throw;
   }
   const bool Synced = msync(Map, (size_t)Off, MS_SYNC) == 0;
   munmap(Map, (size_t)Off);
   close(Fd);
   OMEGA_REQUIRE(Synced, "History: error flushing " + Path);
   return NWritten;
}

} // namespace OMEGA
