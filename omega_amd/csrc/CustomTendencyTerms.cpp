// See CustomTendencyTerms.h
#include "CustomTendencyTerms.h"

#include <cmath>

namespace OMEGA {

ManufacturedSolution::ManufacturedSolution(const HorzMesh *InMesh, R8 WavelengthX, R8 WavelengthY, R8 Amplitude,
                                           bool VelDiff, bool VelHyperDiff, R8 ViscDel2, R8 ViscDel4)
    : Mesh(InMesh) {
   OMEGA_REQUIRE(Mesh && !Mesh->HostOnly, "ManufacturedSolution: needs a device mesh");
   const R8 H0   = Mesh->BottomDepthH(0);              // :76-77 (horizontally uniform resting thickness)
   const R8 Grav = 9.80665, Pii = 3.141592653589793;   // :80-81
   const R8 Kx = 2.0 * Pii / WavelengthX, Ky = 2.0 * Pii / WavelengthY;
   Params.H0 = H0, Params.Eta0 = Amplitude, Params.Kx = Kx, Params.Ky = Ky, Params.Grav = Grav;
   Params.AngFreq = std::sqrt(H0 * Grav * (Kx * Kx + Ky * Ky)); // :84
   Params.VelDiffTendencyEnable = VelDiff ? 1 : 0, Params.VelHyperDiffTendencyEnable = VelHyperDiff ? 1 : 0;
   Params.ViscDel2 = ViscDel2, Params.ViscDel4 = ViscDel4;
   XCell = createDeviceMirrorCopy<Real, 1>("MsXCell", Mesh->XCellH);
   YCell = createDeviceMirrorCopy<Real, 1>("MsYCell", Mesh->YCellH);
   XEdge = createDeviceMirrorCopy<Real, 1>("MsXEdge", Mesh->XEdgeH);
   YEdge = createDeviceMirrorCopy<Real, 1>("MsYEdge", Mesh->YEdgeH);
   FEdge = createDeviceMirrorCopy<Real, 1>("MsFEdge", Mesh->FEdgeH);
}

void ManufacturedSolution::thicknessTendency(const Array2DReal &Tend, R8 T, hipStream_t S) const {
   launchManufacturedThickness(Mesh->NCellsAll, Tend.Ext[1], Tend.Ptr, XCell.Ptr, YCell.Ptr, Params, T, S);
}
void ManufacturedSolution::velocityTendency(const Array2DReal &Tend, R8 T, hipStream_t S) const {
   launchManufacturedVelocity(Mesh->NEdgesAll, Tend.Ext[1], Tend.Ptr, XEdge.Ptr, YEdge.Ptr, FEdge.Ptr,
                              Mesh->AngleEdge.Ptr, Params, T, S);
}

} // namespace OMEGA
