// HorzOperators.h -- the reference's horizontal operator classes (components/omega/src/ocn/HorzOperators.h:9-187,
// HorzOperators.cpp:7-28) with the same names and constructors.  The reference functors are per-element device
// functions a caller wraps in its own parallelFor; here the object's call operator launches the sweep over
// elements [0, N) x all levels as one HIP kernel (kernels/HorzOperators.hip) on the given stream -- what
// test/ocn/HorzOperatorsTest.cpp does around them (parallelFor over NCellsOwned / NEdgesOwned / NVerticesOwned).
#ifndef OMEGA_AMD_HORZOPERATORS_H
#define OMEGA_AMD_HORZOPERATORS_H

#include "HorzMesh.h"
#include "kernels/Kernels.h"

namespace OMEGA {

/// input and output of one operator call must share the row pitch (both made with Array2DReal::levels, or both compact)
inline int samePitch(const Array2DReal &A, const Array2DReal &B) {
   OMEGA_REQUIRE(A.Pitch == B.Pitch && A.Ext[1] == B.Ext[1], "HorzOperators: arrays of different level count / row pitch");
   return A.Pitch;
}

class DivergenceOnCell {
 public:
   explicit DivergenceOnCell(HorzMesh const *Mesh) : Mesh(Mesh) {}
   /// DivCell(i, k) for i < N (N < 0: NCellsAll)
   void operator()(const Array2DReal &DivCell, const Array2DReal &VecEdge, hipStream_t S = nullptr, int N = -1) const {
      launchDivergenceOnCell(Mesh->view(), N < 0 ? Mesh->NCellsAll : N, DivCell.Ext[1], samePitch(DivCell, VecEdge),
                             DivCell.Ptr, VecEdge.Ptr, S);
   }

 private:
   HorzMesh const *Mesh;
};

class GradientOnEdge {
 public:
   explicit GradientOnEdge(HorzMesh const *Mesh) : Mesh(Mesh) {}
   void operator()(const Array2DReal &GradEdge, const Array2DReal &ScalarCell, hipStream_t S = nullptr,
                   int N = -1) const {
      launchGradientOnEdge(Mesh->view(), N < 0 ? Mesh->NEdgesAll : N, GradEdge.Ext[1], samePitch(GradEdge, ScalarCell),
                           GradEdge.Ptr, ScalarCell.Ptr, S);
   }

 private:
   HorzMesh const *Mesh;
};

class CurlOnVertex {
 public:
   explicit CurlOnVertex(HorzMesh const *Mesh) : Mesh(Mesh) {}
   void operator()(const Array2DReal &CurlVertex, const Array2DReal &VecEdge, hipStream_t S = nullptr,
                   int N = -1) const {
      launchCurlOnVertex(Mesh->view(), N < 0 ? Mesh->NVerticesAll : N, CurlVertex.Ext[1], samePitch(CurlVertex, VecEdge),
                         CurlVertex.Ptr, VecEdge.Ptr, S);
   }

 private:
   HorzMesh const *Mesh;
};

class TangentialReconOnEdge {
 public:
   explicit TangentialReconOnEdge(HorzMesh const *Mesh) : Mesh(Mesh) {}
   void operator()(const Array2DReal &ReconEdge, const Array2DReal &VecEdge, hipStream_t S = nullptr,
                   int N = -1) const {
      launchTangentialReconOnEdge(Mesh->view(), N < 0 ? Mesh->NEdgesAll : N, ReconEdge.Ext[1],
                                  samePitch(ReconEdge, VecEdge), ReconEdge.Ptr, VecEdge.Ptr, S);
   }

 private:
   HorzMesh const *Mesh;
};

/// InterpCellToEdge (HorzOperators.h:137-187) on 1-D arrays; the option enum lives in AuxiliaryState.h
class InterpCellToEdge {
 public:
   explicit InterpCellToEdge(HorzMesh const *Mesh) : Mesh(Mesh) {}
   void operator()(const Array1DReal &ArrayEdge, const Array1DReal &ArrayCell, bool Isotropic, hipStream_t S = nullptr,
                   int N = -1) const {
      launchInterpCellToEdge(Mesh->view(), N < 0 ? Mesh->NEdgesAll : N, ArrayEdge.Ptr, ArrayCell.Ptr, Isotropic ? 1 : 0,
                             S);
   }

 private:
   HorzMesh const *Mesh;
};

} // namespace OMEGA
#endif
