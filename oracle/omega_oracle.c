/* omega_oracle.c -- CPU oracle (TEST INFRASTRUCTURE ONLY; see omega_oracle.h).
 *
 * Restates, functor by functor and launch by launch, the reference's
 *   O/src/ocn/HorzOperators.h, O/src/ocn/auxiliaryVars/ *.h, O/src/ocn/TendencyTerms.h,
 *   O/src/ocn/AuxiliaryState.cpp:60-185, O/src/ocn/Tendencies.cpp:257-600,
 *   O/src/ocn/HorzMesh.cpp:527-626, O/src/timeStepping/ *.cpp
 * (O/ = /root/reference/components/omega/) with VecLength = 1 and Real = double.
 * Products are evaluated left to right exactly as written in the reference; compile
 * with -ffp-contract=off.  Loops run over the same index ranges as the reference's
 * parallelFor launches (all local elements incl. halo), element-outer / level-inner.
 */
#include "omega_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define PFOR _Pragma("omp parallel for schedule(static)")

void orc_set_num_threads(int n) {
#ifdef _OPENMP
   omp_set_num_threads(n);
#else
   (void)n;
#endif
}
int orc_get_max_threads(void) {
#ifdef _OPENMP
   return omp_get_max_threads();
#else
   return 1;
#endif
}

/* O/configs/Default.yml:25-52 */
void orc_config_default(orc_config *c) {
   c->ThicknessFluxTendencyEnable   = 1;
   c->PVTendencyEnable              = 1;
   c->KETendencyEnable              = 1;
   c->SSHTendencyEnable             = 1;
   c->VelDiffTendencyEnable         = 1;
   c->VelHyperDiffTendencyEnable    = 1;
   c->WindForcingTendencyEnable     = 0;
   c->BottomDragTendencyEnable      = 0;
   c->TracerHorzAdvTendencyEnable   = 1;
   c->TracerDiffTendencyEnable      = 1;
   c->TracerHyperDiffTendencyEnable = 1;
   c->FluxThicknessUpwind           = 0;
   c->FluxTracerUpwind              = 0;
   c->WindInterpIsotropic           = 1;
   c->ViscDel2                      = 1.0e3;
   c->ViscDel4                      = 1.2e11;
   c->DivFactor                     = 1.0;
   c->EddyDiff2                     = 10.0;
   c->EddyDiff4                     = 0.0;
   c->Density0                      = 1026.0;
   c->BottomDragCoeff               = 0.0;
}

/* HorzMesh::computeEdgeSign / setMasks / setMeshScaling (O/src/ocn/HorzMesh.cpp:527-626).
 * The derived arrays must be zero-filled by the caller (Kokkos zero-initialises views);
 * entries past NEdgesOnCell, and the sentinel rows, keep that zero (EdgeMask's sentinel
 * row keeps the deepCopy value 1). */
void orc_mesh_derive(orc_mesh *m) {
   const int ME = m->MaxEdges, VD = m->VertexDegree, K = m->NVertLayers;
   for (int Cell = 0; Cell < m->NCellsAll; ++Cell) {
      for (int i = 0; i < m->NEdgesOnCell[Cell]; ++i) {
         int Edge = m->EdgesOnCell[Cell * ME + i];
         m->EdgeSignOnCell[Cell * ME + i] =
             (Cell == m->CellsOnEdge[Edge * 2 + 0]) ? -1.0 : 1.0;
      }
   }
   for (int Vertex = 0; Vertex < m->NVerticesAll; ++Vertex) {
      for (int i = 0; i < VD; ++i) {
         int Edge = m->EdgesOnVertex[Vertex * VD + i];
         m->EdgeSignOnVertex[Vertex * VD + i] =
             (Vertex == m->VerticesOnEdge[Edge * 2 + 0]) ? -1.0 : 1.0;
      }
   }
   for (size_t i = 0; i < (size_t)m->NEdgesSize * K; ++i)
      m->EdgeMask[i] = 1.0;
   for (int Edge = 0; Edge < m->NEdgesAll; ++Edge) {
      int Cell1 = m->CellsOnEdge[Edge * 2 + 0], Cell2 = m->CellsOnEdge[Edge * 2 + 1];
      if (!(Cell1 >= 0 && Cell1 < m->NCellsAll) || !(Cell2 >= 0 && Cell2 < m->NCellsAll))
         for (int k = 0; k < K; ++k)
            m->EdgeMask[(size_t)Edge * K + k] = 0.0;
   }
   for (int Edge = 0; Edge < m->NEdgesAll; ++Edge) {
      m->MeshScalingDel2[Edge] = 1.0;
      m->MeshScalingDel4[Edge] = 1.0;
   }
}

#define IX(i, k) ((size_t)(i) * K + (k))
#define IX3(l, i, k, rows) (((size_t)(l) * (rows) + (i)) * K + (k))

/* ======================= HorzOperators.h ======================= */

/* DivergenceOnCell, O/src/ocn/HorzOperators.h:13-33 */
void orc_divergence_on_cell(const orc_mesh *m, int N, double *DivCell, const double *VecEdge) {
   const int K = m->NVertLayers, ME = m->MaxEdges;
   PFOR for (int ICell = 0; ICell < N; ++ICell) {
      const double InvAreaCell = 1. / m->AreaCell[ICell];
      for (int k = 0; k < K; ++k) {
         double DivCellTmp = 0;
         for (int J = 0; J < m->NEdgesOnCell[ICell]; ++J) {
            const int JEdge = m->EdgesOnCell[ICell * ME + J];
            DivCellTmp -= m->DvEdge[JEdge] * m->EdgeSignOnCell[ICell * ME + J] *
                          VecEdge[IX(JEdge, k)] * InvAreaCell;
         }
         DivCell[IX(ICell, k)] = DivCellTmp;
      }
   }
}

/* GradientOnEdge, O/src/ocn/HorzOperators.h:47-60 */
void orc_gradient_on_edge(const orc_mesh *m, int N, double *GradEdge, const double *ScalarCell) {
   const int K = m->NVertLayers;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const double InvDcEdge = 1. / m->DcEdge[IEdge];
      const int JCell0 = m->CellsOnEdge[IEdge * 2 + 0], JCell1 = m->CellsOnEdge[IEdge * 2 + 1];
      for (int k = 0; k < K; ++k)
         GradEdge[IX(IEdge, k)] = InvDcEdge * (ScalarCell[IX(JCell1, k)] - ScalarCell[IX(JCell0, k)]);
   }
}

/* CurlOnVertex, O/src/ocn/HorzOperators.h:71-93 */
void orc_curl_on_vertex(const orc_mesh *m, int N, double *CurlVertex, const double *VecEdge) {
   const int K = m->NVertLayers, VD = m->VertexDegree;
   PFOR for (int IVertex = 0; IVertex < N; ++IVertex) {
      const double InvAreaTriangle = 1. / m->AreaTriangle[IVertex];
      for (int k = 0; k < K; ++k) {
         double CurlVertexTmp = 0;
         for (int J = 0; J < VD; ++J) {
            const int JEdge = m->EdgesOnVertex[IVertex * VD + J];
            CurlVertexTmp += m->DcEdge[JEdge] * m->EdgeSignOnVertex[IVertex * VD + J] *
                             VecEdge[IX(JEdge, k)] * InvAreaTriangle;
         }
         CurlVertex[IX(IVertex, k)] = CurlVertexTmp;
      }
   }
}

/* TangentialReconOnEdge, O/src/ocn/HorzOperators.h:107-126 */
void orc_tangential_recon_on_edge(const orc_mesh *m, int N, double *ReconEdge, const double *VecEdge) {
   const int K = m->NVertLayers, ME2 = m->MaxEdges2;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      for (int k = 0; k < K; ++k) {
         double ReconEdgeTmp = 0;
         for (int J = 0; J < m->NEdgesOnEdge[IEdge]; ++J) {
            const int JEdge = m->EdgesOnEdge[IEdge * ME2 + J];
            ReconEdgeTmp += m->WeightsOnEdge[IEdge * ME2 + J] * VecEdge[IX(JEdge, k)];
         }
         ReconEdge[IX(IEdge, k)] = ReconEdgeTmp;
      }
   }
}

/* InterpCellToEdge, O/src/ocn/HorzOperators.h:137-187 (1-D arrays) */
static double interp_cell_to_edge(const orc_mesh *m, int IEdge, const double *ArrayCell, int Isotropic) {
   if (!Isotropic) { /* interpolateAnisotropic :153-159 */
      const int JCell0 = m->CellsOnEdge[IEdge * 2 + 0], JCell1 = m->CellsOnEdge[IEdge * 2 + 1];
      return 0.5 * (ArrayCell[JCell0] + ArrayCell[JCell1]);
   }
   /* interpolateIsotropic :161-180 */
   const int VD = m->VertexDegree;
   double Accum = 0, AreaAccum = 0;
   for (int J = 0; J < 2; ++J) {
      const int JVertex = m->VerticesOnEdge[IEdge * 2 + J];
      for (int L = 0; L < VD; ++L) {
         const double KiteArea = m->KiteAreasOnVertex[JVertex * VD + L];
         const int LCell       = m->CellsOnVertex[JVertex * VD + L];
         Accum += ArrayCell[LCell] * KiteArea;
         AreaAccum += KiteArea;
      }
   }
   const double InvAreaAccum = 1. / AreaAccum;
   return Accum * InvAreaAccum;
}
void orc_interp_cell_to_edge(const orc_mesh *m, int N, double *OutEdge, const double *ArrayCell, int Isotropic) {
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge)
      OutEdge[IEdge] = interp_cell_to_edge(m, IEdge, ArrayCell, Isotropic);
}

/* ======================= auxiliaryVars ======================= */

/* VorticityAuxVars::computeVarsOnVertex, O/src/ocn/auxiliaryVars/VorticityAuxVars.h:24-59 */
void orc_vorticity_on_vertex(const orc_mesh *m, int N, const orc_aux *a, const double *h, const double *u) {
   const int K = m->NVertLayers, VD = m->VertexDegree;
   PFOR for (int IVertex = 0; IVertex < N; ++IVertex) {
      const double InvAreaTriangle = 1. / m->AreaTriangle[IVertex];
      for (int k = 0; k < K; ++k) {
         double LayerThickVertex = 0, RelVortVertexTmp = 0;
         for (int J = 0; J < VD; ++J) {
            const int JCell = m->CellsOnVertex[IVertex * VD + J];
            const int JEdge = m->EdgesOnVertex[IVertex * VD + J];
            LayerThickVertex += InvAreaTriangle * m->KiteAreasOnVertex[IVertex * VD + J] * h[IX(JCell, k)];
            RelVortVertexTmp += InvAreaTriangle * m->DcEdge[JEdge] *
                                m->EdgeSignOnVertex[IVertex * VD + J] * u[IX(JEdge, k)];
         }
         const double InvLayerThickVertex = 1. / LayerThickVertex;
         a->RelVortVertex[IX(IVertex, k)]        = RelVortVertexTmp;
         a->NormRelVortVertex[IX(IVertex, k)]    = RelVortVertexTmp * InvLayerThickVertex;
         a->NormPlanetVortVertex[IX(IVertex, k)] = m->FVertex[IVertex] * InvLayerThickVertex;
      }
   }
}

/* VorticityAuxVars::computeVarsOnEdge, O/src/ocn/auxiliaryVars/VorticityAuxVars.h:61-76 */
void orc_vorticity_on_edge(const orc_mesh *m, int N, const orc_aux *a) {
   const int K = m->NVertLayers;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const int JVertex0 = m->VerticesOnEdge[IEdge * 2 + 0], JVertex1 = m->VerticesOnEdge[IEdge * 2 + 1];
      for (int k = 0; k < K; ++k) {
         a->NormRelVortEdge[IX(IEdge, k)] =
             0.5 * (a->NormRelVortVertex[IX(JVertex0, k)] + a->NormRelVortVertex[IX(JVertex1, k)]);
         a->NormPlanetVortEdge[IX(IEdge, k)] =
             0.5 * (a->NormPlanetVortVertex[IX(JVertex0, k)] + a->NormPlanetVortVertex[IX(JVertex1, k)]);
      }
   }
}

/* KineticAuxVars::computeVarsOnCell, O/src/ocn/auxiliaryVars/KineticAuxVars.h:20-47 */
void orc_kinetic_on_cell(const orc_mesh *m, int N, const orc_aux *a, const double *u) {
   const int K = m->NVertLayers, ME = m->MaxEdges;
   PFOR for (int ICell = 0; ICell < N; ++ICell) {
      const double InvAreaCell = 1. / m->AreaCell[ICell];
      for (int k = 0; k < K; ++k) {
         double KineticEnergyCellTmp = 0, VelocityDivCellTmp = 0;
         for (int J = 0; J < m->NEdgesOnCell[ICell]; ++J) {
            const int JEdge       = m->EdgesOnCell[ICell * ME + J];
            const double AreaEdge = 0.5 * m->DvEdge[JEdge] * m->DcEdge[JEdge];
            KineticEnergyCellTmp += AreaEdge * 0.5 * InvAreaCell * u[IX(JEdge, k)] * u[IX(JEdge, k)];
            VelocityDivCellTmp -= m->DvEdge[JEdge] * InvAreaCell * m->EdgeSignOnCell[ICell * ME + J] * u[IX(JEdge, k)];
         }
         a->KineticEnergyCell[IX(ICell, k)] = KineticEnergyCellTmp;
         a->VelocityDivCell[IX(ICell, k)]   = VelocityDivCellTmp;
      }
   }
}

/* LayerThicknessAuxVars::computeVarsOnEdge, O/src/ocn/auxiliaryVars/LayerThicknessAuxVars.h:25-61 */
void orc_layerthick_on_edge(const orc_mesh *m, int N, const orc_aux *a, const double *h, const double *u, int Upwind) {
   const int K = m->NVertLayers;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const int JCell0 = m->CellsOnEdge[IEdge * 2 + 0], JCell1 = m->CellsOnEdge[IEdge * 2 + 1];
      for (int k = 0; k < K; ++k) {
         a->MeanLayerThickEdge[IX(IEdge, k)] = 0.5 * (h[IX(JCell0, k)] + h[IX(JCell1, k)]);
         if (!Upwind) {
            a->FluxLayerThickEdge[IX(IEdge, k)] = 0.5 * (h[IX(JCell0, k)] + h[IX(JCell1, k)]);
         } else {
            if (u[IX(IEdge, k)] > 0)
               a->FluxLayerThickEdge[IX(IEdge, k)] = h[IX(JCell0, k)];
            else if (u[IX(IEdge, k)] < 0)
               a->FluxLayerThickEdge[IX(IEdge, k)] = h[IX(JCell1, k)];
            else
               a->FluxLayerThickEdge[IX(IEdge, k)] = fmax(h[IX(JCell0, k)], h[IX(JCell1, k)]);
         }
      }
   }
}

/* LayerThicknessAuxVars::computeVarsOnCells, O/src/ocn/auxiliaryVars/LayerThicknessAuxVars.h:63-82 */
void orc_layerthick_on_cell(const orc_mesh *m, int N, const orc_aux *a, const double *h) {
   const int K = m->NVertLayers;
   PFOR for (int ICell = 0; ICell < N; ++ICell)
      for (int k = 0; k < K; ++k)
         a->SshCell[IX(ICell, k)] = h[IX(ICell, k)] - m->BottomDepth[ICell];
}

/* VelocityDel2AuxVars::computeVarsOnEdge, O/src/ocn/auxiliaryVars/VelocityDel2AuxVars.h:21-45 */
void orc_veldel2_on_edge(const orc_mesh *m, int N, const orc_aux *a, const double *DivCell, const double *RelVortVertex) {
   const int K = m->NVertLayers;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const int JCell0 = m->CellsOnEdge[IEdge * 2 + 0], JCell1 = m->CellsOnEdge[IEdge * 2 + 1];
      const int JVertex0 = m->VerticesOnEdge[IEdge * 2 + 0], JVertex1 = m->VerticesOnEdge[IEdge * 2 + 1];
      const double InvDcEdge = 1. / m->DcEdge[IEdge];
      const double InvDvEdge = 1. / fmax(m->DvEdge[IEdge], 0.25 * m->DcEdge[IEdge]);
      for (int k = 0; k < K; ++k) {
         const double GradDiv  = (DivCell[IX(JCell1, k)] - DivCell[IX(JCell0, k)]) * InvDcEdge;
         const double CurlVort = -(RelVortVertex[IX(JVertex1, k)] - RelVortVertex[IX(JVertex0, k)]) * InvDvEdge;
         a->Del2Edge[IX(IEdge, k)] = m->EdgeMask[IX(IEdge, k)] * GradDiv + CurlVort;
      }
   }
}

/* VelocityDel2AuxVars::computeVarsOnCell, O/src/ocn/auxiliaryVars/VelocityDel2AuxVars.h:47-67 */
void orc_veldel2_on_cell(const orc_mesh *m, int N, const orc_aux *a) {
   const int K = m->NVertLayers, ME = m->MaxEdges;
   PFOR for (int ICell = 0; ICell < N; ++ICell) {
      const double InvAreaCell = 1. / m->AreaCell[ICell];
      for (int k = 0; k < K; ++k) {
         double Del2DivCellTmp = 0;
         for (int J = 0; J < m->NEdgesOnCell[ICell]; ++J) {
            const int JEdge = m->EdgesOnCell[ICell * ME + J];
            Del2DivCellTmp -= m->DvEdge[JEdge] * InvAreaCell * m->EdgeSignOnCell[ICell * ME + J] * a->Del2Edge[IX(JEdge, k)];
         }
         a->Del2DivCell[IX(ICell, k)] = Del2DivCellTmp;
      }
   }
}

/* VelocityDel2AuxVars::computeVarsOnVertex, O/src/ocn/auxiliaryVars/VelocityDel2AuxVars.h:69-89 */
void orc_veldel2_on_vertex(const orc_mesh *m, int N, const orc_aux *a) {
   const int K = m->NVertLayers, VD = m->VertexDegree;
   PFOR for (int IVertex = 0; IVertex < N; ++IVertex) {
      const double InvAreaTriangle = 1. / m->AreaTriangle[IVertex];
      for (int k = 0; k < K; ++k) {
         double Del2RelVortVertexTmp = 0;
         for (int J = 0; J < VD; ++J) {
            const int JEdge = m->EdgesOnVertex[IVertex * VD + J];
            Del2RelVortVertexTmp += InvAreaTriangle * m->DcEdge[JEdge] *
                                    m->EdgeSignOnVertex[IVertex * VD + J] * a->Del2Edge[IX(JEdge, k)];
         }
         a->Del2RelVortVertex[IX(IVertex, k)] = Del2RelVortVertexTmp;
      }
   }
}

/* TracerAuxVars::computeVarsOnEdge, O/src/ocn/auxiliaryVars/TracerAuxVars.h:25-59 */
void orc_tracer_on_edge(const orc_mesh *m, int NT, int N, const orc_aux *a, const double *u, const double *h, const double *tr, int Upwind) {
   const int K = m->NVertLayers, NCS = m->NCellsSize, NES = m->NEdgesSize;
   for (int L = 0; L < NT; ++L) {
      PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
         const int JCell0 = m->CellsOnEdge[IEdge * 2 + 0], JCell1 = m->CellsOnEdge[IEdge * 2 + 1];
         for (int k = 0; k < K; ++k) {
            const double ht0 = h[IX(JCell0, k)] * tr[IX3(L, JCell0, k, NCS)];
            const double ht1 = h[IX(JCell1, k)] * tr[IX3(L, JCell1, k, NCS)];
            double r;
            if (!Upwind)
               r = 0.5 * (ht0 + ht1);
            else if (u[IX(IEdge, k)] > 0)
               r = ht0;
            else if (u[IX(IEdge, k)] < 0)
               r = ht1;
            else
               r = fmax(ht0, ht1);
            a->HTracersEdge[IX3(L, IEdge, k, NES)] = r;
         }
      }
   }
}

/* TracerAuxVars::computeVarsOnCells, O/src/ocn/auxiliaryVars/TracerAuxVars.h:61-91 */
void orc_tracer_on_cell(const orc_mesh *m, int NT, int N, const orc_aux *a, const double *hMeanEdge, const double *tr) {
   const int K = m->NVertLayers, ME = m->MaxEdges, NCS = m->NCellsSize;
   for (int L = 0; L < NT; ++L) {
      PFOR for (int ICell = 0; ICell < N; ++ICell) {
         const double InvAreaCell = 1. / m->AreaCell[ICell];
         for (int k = 0; k < K; ++k) {
            double Del2TrCellTmp = 0;
            for (int J = 0; J < m->NEdgesOnCell[ICell]; ++J) {
               const int JEdge  = m->EdgesOnCell[ICell * ME + J];
               const int JCell0 = m->CellsOnEdge[JEdge * 2 + 0], JCell1 = m->CellsOnEdge[JEdge * 2 + 1];
               const double DvDcEdge   = m->DvEdge[JEdge] / m->DcEdge[JEdge];
               const double TracerGrad = tr[IX3(L, JCell1, k, NCS)] - tr[IX3(L, JCell0, k, NCS)];
               Del2TrCellTmp -= m->EdgeMask[IX(JEdge, k)] * m->EdgeSignOnCell[ICell * ME + J] * DvDcEdge *
                                hMeanEdge[IX(JEdge, k)] * TracerGrad;
            }
            a->Del2TracersCell[IX3(L, ICell, k, NCS)] = Del2TrCellTmp * InvAreaCell;
         }
      }
   }
}

/* WindForcingAuxVars::computeVarsOnEdge, O/src/ocn/auxiliaryVars/WindForcingAuxVars.h:22-29 */
void orc_wind_on_edge(const orc_mesh *m, int N, const orc_aux *a, int Isotropic) {
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const double ZonalStressEdge = interp_cell_to_edge(m, IEdge, a->ZonalStressCell, Isotropic);
      const double MeridStressEdge = interp_cell_to_edge(m, IEdge, a->MeridStressCell, Isotropic);
      a->NormalStressEdge[IEdge] =
          cos(m->AngleEdge[IEdge]) * ZonalStressEdge + sin(m->AngleEdge[IEdge]) * MeridStressEdge;
   }
}

/* AuxiliaryState::computeMomAux, O/src/ocn/AuxiliaryState.cpp:60-143 (launch order kept) */
void orc_aux_compute_mom_aux(const orc_mesh *m, const orc_config *c, const orc_aux *a, const double *h, const double *u) {
   orc_vorticity_on_vertex(m, m->NVerticesAll, a, h, u);            /* vertexAuxState1 :79-85  */
   orc_kinetic_on_cell(m, m->NCellsAll, a, u);                      /* cellAuxState1   :88-93  */
   orc_wind_on_edge(m, m->NEdgesAll, a, c->WindInterpIsotropic);    /* edgeAuxState1   :99-103 */
   orc_vorticity_on_edge(m, m->NEdgesAll, a);                       /* edgeAuxState2   :106-115 */
   orc_layerthick_on_edge(m, m->NEdgesAll, a, h, u, c->FluxThicknessUpwind);
   orc_veldel2_on_edge(m, m->NEdgesAll, a, a->VelocityDivCell, a->RelVortVertex);
   orc_veldel2_on_vertex(m, m->NVerticesAll, a);                    /* vertexAuxState2 :118-123 */
   orc_veldel2_on_cell(m, m->NCellsAll, a);                         /* cellAuxState2   :126-131 */
   orc_layerthick_on_cell(m, m->NCellsAll, a, h);                   /* cellAuxState3   :134-140 */
}

/* AuxiliaryState::computeAll, O/src/ocn/AuxiliaryState.cpp:146-185 */
void orc_aux_compute_all(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, const double *h, const double *u, const double *tr) {
   orc_aux_compute_mom_aux(m, c, a, h, u);
   orc_tracer_on_edge(m, NT, m->NEdgesAll, a, u, h, tr, c->FluxTracerUpwind); /* edgeAuxState4 :165-171 */
   orc_tracer_on_cell(m, NT, m->NCellsAll, a, a->MeanLayerThickEdge, tr);     /* cellAuxState4 :176-182 */
}

/* ======================= TendencyTerms.h ======================= */

/* ThicknessFluxDivOnCell, O/src/ocn/TendencyTerms.h:35-58 */
void orc_thickness_flux_div_on_cell(const orc_mesh *m, int N, double *Tend, const double *ThicknessFlux, const double *NormalVelEdge) {
   const int K = m->NVertLayers, ME = m->MaxEdges;
   PFOR for (int ICell = 0; ICell < N; ++ICell) {
      const double InvAreaCell = 1. / m->AreaCell[ICell];
      for (int k = 0; k < K; ++k) {
         double DivTmp = 0;
         for (int J = 0; J < m->NEdgesOnCell[ICell]; ++J) {
            const int JEdge = m->EdgesOnCell[ICell * ME + J];
            DivTmp -= m->DvEdge[JEdge] * m->EdgeSignOnCell[ICell * ME + J] * ThicknessFlux[IX(JEdge, k)] *
                      NormalVelEdge[IX(JEdge, k)] * InvAreaCell;
         }
         Tend[IX(ICell, k)] -= DivTmp;
      }
   }
}

/* PotentialVortHAdvOnEdge, O/src/ocn/TendencyTerms.h:81-108 */
void orc_pv_hadv_on_edge(const orc_mesh *m, int N, double *Tend, const double *NormRVortEdge, const double *NormFEdge,
                         const double *FluxLayerThickEdge, const double *NormVelEdge) {
   const int K = m->NVertLayers, ME2 = m->MaxEdges2;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      for (int k = 0; k < K; ++k) {
         double VortTmp = 0;
         for (int J = 0; J < m->NEdgesOnEdge[IEdge]; ++J) {
            const int JEdge = m->EdgesOnEdge[IEdge * ME2 + J];
            const double NormVort = (NormRVortEdge[IX(IEdge, k)] + NormFEdge[IX(IEdge, k)] +
                                     NormRVortEdge[IX(JEdge, k)] + NormFEdge[IX(JEdge, k)]) * 0.5;
            VortTmp += m->WeightsOnEdge[IEdge * ME2 + J] * FluxLayerThickEdge[IX(JEdge, k)] *
                       NormVelEdge[IX(JEdge, k)] * NormVort;
         }
         Tend[IX(IEdge, k)] += m->EdgeMask[IX(IEdge, k)] * VortTmp;
      }
   }
}

/* KEGradOnEdge, O/src/ocn/TendencyTerms.h:127-140 */
void orc_ke_grad_on_edge(const orc_mesh *m, int N, double *Tend, const double *KECell) {
   const int K = m->NVertLayers;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const int JCell0 = m->CellsOnEdge[IEdge * 2 + 0], JCell1 = m->CellsOnEdge[IEdge * 2 + 1];
      const double InvDcEdge = 1. / m->DcEdge[IEdge];
      for (int k = 0; k < K; ++k)
         Tend[IX(IEdge, k)] -= m->EdgeMask[IX(IEdge, k)] * (KECell[IX(JCell1, k)] - KECell[IX(JCell0, k)]) * InvDcEdge;
   }
}

/* SSHGradOnEdge, O/src/ocn/TendencyTerms.h:159-173 (Grav :176) */
void orc_ssh_grad_on_edge(const orc_mesh *m, int N, double *Tend, const double *SshCell) {
   const int K = m->NVertLayers;
   const double Grav = 9.80665;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const int ICell0 = m->CellsOnEdge[IEdge * 2 + 0], ICell1 = m->CellsOnEdge[IEdge * 2 + 1];
      const double InvDcEdge = 1. / m->DcEdge[IEdge];
      for (int k = 0; k < K; ++k)
         Tend[IX(IEdge, k)] -= m->EdgeMask[IX(IEdge, k)] * Grav * (SshCell[IX(ICell1, k)] - SshCell[IX(ICell0, k)]) * InvDcEdge;
   }
}

/* VelocityDiffusionOnEdge, O/src/ocn/TendencyTerms.h:195-219 */
void orc_velocity_diffusion_on_edge(const orc_mesh *m, int N, double *Tend, const double *DivCell, const double *RVortVertex, double ViscDel2) {
   const int K = m->NVertLayers;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const int ICell0 = m->CellsOnEdge[IEdge * 2 + 0], ICell1 = m->CellsOnEdge[IEdge * 2 + 1];
      const int IVertex0 = m->VerticesOnEdge[IEdge * 2 + 0], IVertex1 = m->VerticesOnEdge[IEdge * 2 + 1];
      const double DcEdgeInv = 1. / m->DcEdge[IEdge];
      const double DvEdgeInv = 1. / m->DvEdge[IEdge];
      for (int k = 0; k < K; ++k) {
         const double Del2U = ((DivCell[IX(ICell1, k)] - DivCell[IX(ICell0, k)]) * DcEdgeInv -
                               (RVortVertex[IX(IVertex1, k)] - RVortVertex[IX(IVertex0, k)]) * DvEdgeInv);
         Tend[IX(IEdge, k)] += m->EdgeMask[IX(IEdge, k)] * ViscDel2 * m->MeshScalingDel2[IEdge] * Del2U;
      }
   }
}

/* VelocityHyperDiffOnEdge, O/src/ocn/TendencyTerms.h:244-269 */
void orc_velocity_hyperdiff_on_edge(const orc_mesh *m, int N, double *Tend, const double *Del2DivCell, const double *Del2RVortVertex,
                                    double ViscDel4, double DivFactor) {
   const int K = m->NVertLayers;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const int ICell0 = m->CellsOnEdge[IEdge * 2 + 0], ICell1 = m->CellsOnEdge[IEdge * 2 + 1];
      const int IVertex0 = m->VerticesOnEdge[IEdge * 2 + 0], IVertex1 = m->VerticesOnEdge[IEdge * 2 + 1];
      const double DcEdgeInv = 1. / m->DcEdge[IEdge];
      const double DvEdgeInv = 1. / m->DvEdge[IEdge];
      for (int k = 0; k < K; ++k) {
         const double Del2U = (DivFactor * (Del2DivCell[IX(ICell1, k)] - Del2DivCell[IX(ICell0, k)]) * DcEdgeInv -
                               (Del2RVortVertex[IX(IVertex1, k)] - Del2RVortVertex[IX(IVertex0, k)]) * DvEdgeInv);
         Tend[IX(IEdge, k)] -= m->EdgeMask[IX(IEdge, k)] * ViscDel4 * m->MeshScalingDel4[IEdge] * Del2U;
      }
   }
}

/* WindForcingOnEdge, O/src/ocn/TendencyTerms.h:291-301 (acts at K = 0 only) */
void orc_wind_forcing_on_edge(const orc_mesh *m, int N, double *Tend, const double *NormalStressEdge, const double *LayerThickEdge, double SaltWaterDensity) {
   const int K = m->NVertLayers;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const int k = 0;
      const double InvThickEdge = 1. / LayerThickEdge[IX(IEdge, k)];
      Tend[IX(IEdge, k)] += m->EdgeMask[IX(IEdge, k)] * InvThickEdge * NormalStressEdge[IEdge] / SaltWaterDensity;
   }
}

/* BottomDragOnEdge, O/src/ocn/TendencyTerms.h:319-334 */
void orc_bottom_drag_on_edge(const orc_mesh *m, int N, double *Tend, const double *NormalVelEdge, const double *KECell, const double *LayerThickEdge, double Coeff) {
   const int K = m->NVertLayers;
   const int KBot = K - 1;
   PFOR for (int IEdge = 0; IEdge < N; ++IEdge) {
      const int JCell0 = m->CellsOnEdge[IEdge * 2 + 0], JCell1 = m->CellsOnEdge[IEdge * 2 + 1];
      const double VelNormEdge  = sqrt(KECell[IX(JCell0, KBot)] + KECell[IX(JCell1, KBot)]);
      const double InvThickEdge = 1. / LayerThickEdge[IX(IEdge, KBot)];
      Tend[IX(IEdge, KBot)] -= m->EdgeMask[IX(IEdge, KBot)] * Coeff * VelNormEdge * InvThickEdge * NormalVelEdge[IX(IEdge, KBot)];
   }
}

/* TracerHorzAdvOnCell, O/src/ocn/TendencyTerms.h:349-373 */
void orc_tracer_horz_adv_on_cell(const orc_mesh *m, int NT, int N, double *Tend, int TendRows, const double *NormVelEdge, const double *HTracersOnEdge) {
   const int K = m->NVertLayers, ME = m->MaxEdges, NES = m->NEdgesSize;
   for (int L = 0; L < NT; ++L) {
      PFOR for (int ICell = 0; ICell < N; ++ICell) {
         const double InvAreaCell = 1. / m->AreaCell[ICell];
         for (int k = 0; k < K; ++k) {
            double HAdvTmp = 0;
            for (int J = 0; J < m->NEdgesOnCell[ICell]; ++J) {
               const int JEdge = m->EdgesOnCell[ICell * ME + J];
               HAdvTmp -= m->EdgeMask[IX(JEdge, k)] * m->DvEdge[JEdge] * m->EdgeSignOnCell[ICell * ME + J] *
                          HTracersOnEdge[IX3(L, JEdge, k, NES)] * NormVelEdge[IX(JEdge, k)] * InvAreaCell;
            }
            Tend[IX3(L, ICell, k, TendRows)] -= HAdvTmp;
         }
      }
   }
}

/* TracerDiffOnCell, O/src/ocn/TendencyTerms.h:394-426 */
void orc_tracer_diff_on_cell(const orc_mesh *m, int NT, int N, double *Tend, int TendRows, const double *TracerCell, const double *MeanLayerThickEdge, double EddyDiff2) {
   const int K = m->NVertLayers, ME = m->MaxEdges, NCS = m->NCellsSize;
   for (int L = 0; L < NT; ++L) {
      PFOR for (int ICell = 0; ICell < N; ++ICell) {
         const double InvAreaCell = 1. / m->AreaCell[ICell];
         for (int k = 0; k < K; ++k) {
            double DiffTmp = 0;
            for (int J = 0; J < m->NEdgesOnCell[ICell]; ++J) {
               const int JEdge  = m->EdgesOnCell[ICell * ME + J];
               const int JCell0 = m->CellsOnEdge[JEdge * 2 + 0], JCell1 = m->CellsOnEdge[JEdge * 2 + 1];
               const double RTemp      = m->MeshScalingDel2[JEdge] * m->DvEdge[JEdge] / m->DcEdge[JEdge];
               const double TracerGrad = (TracerCell[IX3(L, JCell1, k, NCS)] - TracerCell[IX3(L, JCell0, k, NCS)]);
               DiffTmp -= m->EdgeMask[IX(JEdge, k)] * m->EdgeSignOnCell[ICell * ME + J] * RTemp *
                          MeanLayerThickEdge[IX(JEdge, k)] * TracerGrad;
            }
            Tend[IX3(L, ICell, k, TendRows)] += EddyDiff2 * DiffTmp * InvAreaCell;
         }
      }
   }
}

/* TracerHyperDiffOnCell, O/src/ocn/TendencyTerms.h:449-480 */
void orc_tracer_hyperdiff_on_cell(const orc_mesh *m, int NT, int N, double *Tend, int TendRows, const double *TrDel2Cell, double EddyDiff4) {
   const int K = m->NVertLayers, ME = m->MaxEdges, NCS = m->NCellsSize;
   for (int L = 0; L < NT; ++L) {
      PFOR for (int ICell = 0; ICell < N; ++ICell) {
         const double InvAreaCell = 1. / m->AreaCell[ICell];
         for (int k = 0; k < K; ++k) {
            double HypTmp = 0;
            for (int J = 0; J < m->NEdgesOnCell[ICell]; ++J) {
               const int JEdge  = m->EdgesOnCell[ICell * ME + J];
               const int JCell0 = m->CellsOnEdge[JEdge * 2 + 0], JCell1 = m->CellsOnEdge[JEdge * 2 + 1];
               const double RTemp      = m->MeshScalingDel4[JEdge] * m->DvEdge[JEdge] / m->DcEdge[JEdge];
               const double Del2TrGrad = (TrDel2Cell[IX3(L, JCell1, k, NCS)] - TrDel2Cell[IX3(L, JCell0, k, NCS)]);
               HypTmp -= m->EdgeMask[IX(JEdge, k)] * m->EdgeSignOnCell[ICell * ME + J] * RTemp * Del2TrGrad;
            }
            Tend[IX3(L, ICell, k, TendRows)] -= EddyDiff4 * HypTmp * InvAreaCell;
         }
      }
   }
}

/* ======================= Tendencies.cpp ======================= */

static void fill0(double *a, size_t n) { memset(a, 0, n * sizeof(double)); }

/* ---- ManufacturedSolution, O/src/ocn/CustomTendencyTerms.cpp ---- */
static const orc_manufactured *g_custom = NULL;
static double g_time = 0.0, g_sim_time = 0.0;
void orc_set_custom_tendency(const orc_manufactured *ms) { g_custom = ms; }
/* DecayVelocityTendency of the reference's time-stepper test (O/test/timeStepping/TimeStepperTest.cpp:49-73):
 * NormalVelTend(IEdge, K) -= Coeff * NormalVelEdge(IEdge, K) as the custom velocity tendency */
static int g_decay_on = 0;
static double g_decay_coeff = 0.5;
void orc_set_decay_velocity_tendency(int On, double Coeff) { g_decay_on = On, g_decay_coeff = Coeff; }
void orc_set_time(double t) { g_time = t; }
void orc_set_sim_time(double t) { g_sim_time = t, g_time = t; }

/* ManufacturedSolution::init, CustomTendencyTerms.cpp:18-107 */
void orc_manufactured_init(orc_manufactured *ms, const orc_mesh *m, const orc_config *c, double WavelengthX,
                           double WavelengthY, double Amplitude) {
   const double H0 = m->BottomDepth[0]; /* :76-77 */
   const double Grav = 9.80665, Pii = 3.141592653589793;
   const double Kx = 2.0 * Pii / WavelengthX, Ky = 2.0 * Pii / WavelengthY;
   ms->H0 = H0, ms->Eta0 = Amplitude, ms->Kx = Kx, ms->Ky = Ky, ms->Grav = Grav;
   ms->AngFreq = sqrt(H0 * Grav * (Kx * Kx + Ky * Ky)); /* :84 */
   ms->VelDiffTendencyEnable = c->VelDiffTendencyEnable, ms->VelHyperDiffTendencyEnable = c->VelHyperDiffTendencyEnable;
   ms->ViscDel2 = c->ViscDel2, ms->ViscDel4 = c->ViscDel4;
}

/* ManufacturedThicknessTendency::operator(), CustomTendencyTerms.cpp:112-145 */
void orc_manufactured_thickness_tend(const orc_mesh *m, const orc_manufactured *ms, double *hTend, double T) {
   const int K = m->NVertLayers;
#pragma omp parallel for
   for (int ICell = 0; ICell < m->NCellsAll; ++ICell) {
      const double X = ms->XCell[ICell], Y = ms->YCell[ICell];
      const double Phase = ms->Kx * X + ms->Ky * Y - ms->AngFreq * T;
      for (int k = 0; k < K; ++k)
         hTend[(size_t)ICell * K + k] += ms->Eta0 * (-ms->H0 * (ms->Kx + ms->Ky) * sin(Phase) - ms->AngFreq * cos(Phase) +
                                                     ms->Eta0 * (ms->Kx + ms->Ky) * cos(2.0 * Phase));
   }
}

/* ManufacturedVelocityTendency::operator(), CustomTendencyTerms.cpp:150-208 */
void orc_manufactured_velocity_tend(const orc_mesh *m, const orc_manufactured *ms, double *uTend, double T) {
   const int K = m->NVertLayers;
   const double Kx2 = ms->Kx * ms->Kx, Ky2 = ms->Ky * ms->Ky, Kx4 = Kx2 * Kx2, Ky4 = Ky2 * Ky2;
#pragma omp parallel for
   for (int IEdge = 0; IEdge < m->NEdgesAll; ++IEdge) {
      const double X = ms->XEdge[IEdge], Y = ms->YEdge[IEdge];
      const double Phase       = ms->Kx * X + ms->Ky * Y - ms->AngFreq * T;
      const double SourceTerm0 = ms->AngFreq * sin(Phase) - 0.5 * ms->Eta0 * (ms->Kx + ms->Ky) * sin(2.0 * Phase);
      double U = ms->Eta0 * ((-ms->FEdge[IEdge] + ms->Grav * ms->Kx) * cos(Phase) + SourceTerm0);
      double V = ms->Eta0 * ((ms->FEdge[IEdge] + ms->Grav * ms->Ky) * cos(Phase) + SourceTerm0);
      if (ms->VelDiffTendencyEnable) {
         U += ms->ViscDel2 * ms->Eta0 * (Kx2 + Ky2) * cos(Phase);
         V += ms->ViscDel2 * ms->Eta0 * (Kx2 + Ky2) * cos(Phase);
      }
      if (ms->VelHyperDiffTendencyEnable) {
         U -= ms->ViscDel4 * ms->Eta0 * ((Kx4 + Ky4 + Kx2 * Ky2) * cos(Phase));
         V -= ms->ViscDel4 * ms->Eta0 * ((Kx4 + Ky4 + Kx2 * Ky2) * cos(Phase));
      }
      const double Src = cos(m->AngleEdge[IEdge]) * U + sin(m->AngleEdge[IEdge]) * V;
      for (int k = 0; k < K; ++k)
         uTend[(size_t)IEdge * K + k] += Src;
   }
}

/* Tendencies::computeThicknessTendenciesOnly, O/src/ocn/Tendencies.cpp:257-297 */
void orc_tend_thickness_only(const orc_mesh *m, const orc_config *c, const orc_aux *a, double *hTend, const double *u) {
   fill0(hTend, (size_t)m->NCellsSize * m->NVertLayers); /* deepCopy(...,0) :272 */
   if (c->ThicknessFluxTendencyEnable)
      orc_thickness_flux_div_on_cell(m, m->NCellsAll, hTend, a->FluxLayerThickEdge, u);
   if (g_custom) /* CustomThicknessTend :288-291 */
      orc_manufactured_thickness_tend(m, g_custom, hTend, g_time);
}

/* Tendencies::computeVelocityTendenciesOnly, O/src/ocn/Tendencies.cpp:301-423 */
void orc_tend_velocity_only(const orc_mesh *m, const orc_config *c, const orc_aux *a, double *uTend, const double *u) {
   const int N = m->NEdgesAll;
   fill0(uTend, (size_t)m->NEdgesSize * m->NVertLayers); /* :320 */
   if (c->PVTendencyEnable)
      orc_pv_hadv_on_edge(m, N, uTend, a->NormRelVortEdge, a->NormPlanetVortEdge, a->FluxLayerThickEdge, u);
   if (c->KETendencyEnable)
      orc_ke_grad_on_edge(m, N, uTend, a->KineticEnergyCell);
   if (c->SSHTendencyEnable)
      orc_ssh_grad_on_edge(m, N, uTend, a->SshCell);
   if (c->VelDiffTendencyEnable)
      orc_velocity_diffusion_on_edge(m, N, uTend, a->VelocityDivCell, a->RelVortVertex, c->ViscDel2);
   if (c->VelHyperDiffTendencyEnable)
      orc_velocity_hyperdiff_on_edge(m, N, uTend, a->Del2DivCell, a->Del2RelVortVertex, c->ViscDel4, c->DivFactor);
   if (c->WindForcingTendencyEnable)
      orc_wind_forcing_on_edge(m, N, uTend, a->NormalStressEdge, a->MeanLayerThickEdge, c->Density0);
   if (c->BottomDragTendencyEnable)
      orc_bottom_drag_on_edge(m, N, uTend, u, a->KineticEnergyCell, a->MeanLayerThickEdge, c->BottomDragCoeff);
   if (g_custom) /* CustomVelocityTend :416-419 */
      orc_manufactured_velocity_tend(m, g_custom, uTend, g_time);
   if (g_decay_on) { /* CustomVelocityTend = DecayVelocityTendency, TimeStepperTest.cpp:66-72 */
      const int K = m->NVertLayers;
#pragma omp parallel for
      for (int IEdge = 0; IEdge < N; ++IEdge)
         for (int k = 0; k < K; ++k)
            uTend[(size_t)IEdge * K + k] -= g_decay_coeff * u[(size_t)IEdge * K + k];
   }
}

/* Tendencies::computeTracerTendenciesOnly, O/src/ocn/Tendencies.cpp:427-486 */
void orc_tend_tracer_only(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, double *trTend, const double *u, const double *tr) {
   const int N = m->NCellsAll, R = m->NCellsSize;
   fill0(trTend, (size_t)NT * m->NCellsSize * m->NVertLayers); /* :442 */
   if (c->TracerHorzAdvTendencyEnable)
      orc_tracer_horz_adv_on_cell(m, NT, N, trTend, R, u, a->HTracersEdge);
   if (c->TracerDiffTendencyEnable)
      orc_tracer_diff_on_cell(m, NT, N, trTend, R, tr, a->MeanLayerThickEdge, c->EddyDiff2);
   if (c->TracerHyperDiffTendencyEnable)
      orc_tracer_hyperdiff_on_cell(m, NT, N, trTend, R, a->Del2TracersCell, c->EddyDiff4);
}

/* Tendencies::computeThicknessTendencies, O/src/ocn/Tendencies.cpp:488-519 */
void orc_tend_compute_thickness(const orc_mesh *m, const orc_config *c, const orc_aux *a, double *hTend, const double *h, const double *u) {
   orc_layerthick_on_edge(m, m->NEdgesAll, a, h, u, c->FluxThicknessUpwind);
   orc_tend_thickness_only(m, c, a, hTend, u);
}

/* Tendencies::computeVelocityTendencies, O/src/ocn/Tendencies.cpp:521-535 */
void orc_tend_compute_velocity(const orc_mesh *m, const orc_config *c, const orc_aux *a, double *uTend, const double *h, const double *u) {
   orc_aux_compute_mom_aux(m, c, a, h, u);
   orc_tend_velocity_only(m, c, a, uTend, u);
}

/* Tendencies::computeTracerTendencies, O/src/ocn/Tendencies.cpp:537-575 */
void orc_tend_compute_tracer(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, double *trTend, const double *h, const double *u, const double *tr) {
   orc_tracer_on_edge(m, NT, m->NEdgesAll, a, u, h, tr, c->FluxTracerUpwind);
   orc_tracer_on_cell(m, NT, m->NCellsAll, a, a->MeanLayerThickEdge, tr);
   orc_tend_tracer_only(m, c, a, NT, trTend, u, tr);
}

/* Tendencies::computeAllTendencies, O/src/ocn/Tendencies.cpp:579-600 */
void orc_tend_compute_all(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, double *hTend, double *uTend, double *trTend,
                          const double *h, const double *u, const double *tr) {
   orc_aux_compute_all(m, c, a, NT, h, u, tr);
   orc_tend_thickness_only(m, c, a, hTend, u);
   orc_tend_velocity_only(m, c, a, uTend, u);
   orc_tend_tracer_only(m, c, a, NT, trTend, u, tr);
}

/* ======================= TimeStepper.cpp ======================= */

/* TimeStepper::updateThicknessByTend, O/src/timeStepping/TimeStepper.cpp:378-401 */
void orc_update_thickness_by_tend(const orc_mesh *m, double *h1, const double *h2, const double *hTend, double CoeffSeconds) {
   const int K = m->NVertLayers;
   PFOR for (int ICell = 0; ICell < m->NCellsAll; ++ICell)
      for (int k = 0; k < K; ++k)
         h1[IX(ICell, k)] = h2[IX(ICell, k)] + CoeffSeconds * hTend[IX(ICell, k)];
}

/* TimeStepper::updateVelocityByTend, O/src/timeStepping/TimeStepper.cpp:407-430 */
void orc_update_velocity_by_tend(const orc_mesh *m, double *u1, const double *u2, const double *uTend, double CoeffSeconds) {
   const int K = m->NVertLayers;
   PFOR for (int IEdge = 0; IEdge < m->NEdgesAll; ++IEdge)
      for (int k = 0; k < K; ++k)
         u1[IX(IEdge, k)] = u2[IX(IEdge, k)] + CoeffSeconds * uTend[IX(IEdge, k)];
}

/* TimeStepper::updateTracersByTend, O/src/timeStepping/TimeStepper.cpp:447-469 */
void orc_update_tracers_by_tend(const orc_mesh *m, int NT, double *NextTr, const double *CurTr, const double *h1, const double *h2,
                                const double *trTend, double CoeffSeconds) {
   const int K = m->NVertLayers, NCS = m->NCellsSize;
   for (int L = 0; L < NT; ++L) {
      PFOR for (int ICell = 0; ICell < m->NCellsAll; ++ICell)
         for (int k = 0; k < K; ++k)
            NextTr[IX3(L, ICell, k, NCS)] =
                (CurTr[IX3(L, ICell, k, NCS)] * h2[IX(ICell, k)] + CoeffSeconds * trTend[IX3(L, ICell, k, NCS)]) / h1[IX(ICell, k)];
   }
}

/* TimeStepper::weightTracers, O/src/timeStepping/TimeStepper.cpp:473-487 */
void orc_weight_tracers(const orc_mesh *m, int NT, double *NextTr, const double *CurTr, const double *hCur) {
   const int K = m->NVertLayers, NCS = m->NCellsSize;
   for (int L = 0; L < NT; ++L) {
      PFOR for (int ICell = 0; ICell < m->NCellsAll; ++ICell)
         for (int k = 0; k < K; ++k)
            NextTr[IX3(L, ICell, k, NCS)] = CurTr[IX3(L, ICell, k, NCS)] * hCur[IX(ICell, k)];
   }
}

/* TimeStepper::accumulateTracersUpdate, O/src/timeStepping/TimeStepper.cpp:492-507 */
void orc_accumulate_tracers_update(const orc_mesh *m, int NT, double *AccumTr, const double *trTend, double CoeffSeconds) {
   const int K = m->NVertLayers, NCS = m->NCellsSize;
   for (int L = 0; L < NT; ++L) {
      PFOR for (int ICell = 0; ICell < m->NCellsAll; ++ICell)
         for (int k = 0; k < K; ++k)
            AccumTr[IX3(L, ICell, k, NCS)] += CoeffSeconds * trTend[IX3(L, ICell, k, NCS)];
   }
}

/* TimeStepper::finalizeTracersUpdate, O/src/timeStepping/TimeStepper.cpp:511-524 */
void orc_finalize_tracers_update(const orc_mesh *m, int NT, double *NextTr, const double *hNext) {
   const int K = m->NVertLayers, NCS = m->NCellsSize;
   for (int L = 0; L < NT; ++L) {
      PFOR for (int ICell = 0; ICell < m->NCellsAll; ++ICell)
         for (int k = 0; k < K; ++k)
            NextTr[IX3(L, ICell, k, NCS)] /= hNext[IX(ICell, k)];
   }
}

/* ---- TimeFrac arithmetic behind `Real * TimeInterval` and TimeInterval::get(seconds) ---- */
typedef struct {
   long long Whole, Numer, Denom;
} tfrac;

static long long tf_gcd(long long a, long long b) { /* TimeFracGCD */
   a = llabs(a);
   b = llabs(b);
   if (a == 0)
      return b ? b : 1;
   if (b == 0)
      return a;
   while (b) {
      long long t = a % b;
      a = b;
      b = t;
   }
   return a;
}

/* TimeFrac::simplify, O/src/infra/TimeMgr.cpp:956-1000 */
static void tf_simplify(tfrac *f) {
   long long W;
   if (llabs((W = f->Numer / f->Denom)) >= 1) {
      f->Whole += W;
      f->Numer %= f->Denom;
   }
   if (f->Whole > 0 && ((f->Numer < 0 && f->Denom > 0) || (f->Denom < 0 && f->Numer > 0))) {
      f->Whole--;
      f->Numer += f->Denom;
   } else if ((f->Whole < 0 && (f->Numer > 0 && f->Denom > 0)) || (f->Denom < 0 && f->Numer < 0)) {
      f->Whole++;
      f->Numer -= f->Denom;
   }
   if (f->Denom < 0) {
      f->Denom *= -1;
      f->Numer *= -1;
   }
   long long G = tf_gcd(f->Numer, f->Denom);
   f->Numer /= G;
   f->Denom /= G;
}

/* TimeFrac::setSeconds, O/src/infra/TimeMgr.cpp:193-283 (continued fractions) */
static tfrac tf_set_seconds(double Seconds) {
   tfrac f = {0, 0, 1};
   double Rabs = fabs(Seconds);
   int Sign = (Seconds < 0) ? -1 : 1;
   double Target = Rabs;
   if (Target == 0.0)
      return f;
   if (Target >= 1.0) {
      long long W = (long long)Rabs;
      Target -= (double)W;
      f.Whole = Sign * W;
      if (Target < 1e-17)
         return f;
   }
   double P = pow(10.0, -(DBL_DIG - (int)log10(Rabs)));
   double R = Target;
   long long Nprevprev = 0, Nprev = 1, Dprevprev = 1, Dprev = 0, A, N, D;
   double F = 0.0;
   for (;;) {
      A = (long long)R;
      N = A * Nprev + Nprevprev;
      D = A * Dprev + Dprevprev;
      if (fabs((double)N / (double)D - Target) < P)
         break;
      F = R - (double)A;
      if (F < 1e-17)
         break;
      R = 1.0 / F;
      Nprevprev = Nprev;
      Nprev = N;
      Dprevprev = Dprev;
      Dprev = D;
   }
   f.Numer = N * Sign;
   f.Denom = D;
   tf_simplify(&f);
   return f;
}

/* TimeFrac::operator*(R8) :747-767 then TimeFrac::getSeconds :382-391 */
double orc_coeff_seconds(double Mult, double TimeStepSeconds) {
   tfrac T = tf_set_seconds(TimeStepSeconds);
   tfrac M = tf_set_seconds(Mult);
   tfrac P = {0, 0, 1};
   P.Denom = T.Denom * M.Denom;
   P.Numer = (T.Whole * T.Denom + T.Numer) * (M.Whole * M.Denom + M.Numer);
   tf_simplify(&P);
   return (double)P.Whole + (double)P.Numer / (double)P.Denom;
}

/* RungeKutta4Stepper::doStep, O/src/timeStepping/RungeKutta4Stepper.cpp:68-137; coefficients :25-38 */
void orc_rk4_step(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, orc_state *s, double dt, orc_exchange_fn ex, void *ctx) {
   const double RKA[4] = {0, 1. / 2, 1. / 2, 1};
   const double RKB[4] = {1. / 6, 1. / 3, 1. / 3, 1. / 6};
   const double RKC[4] = {0, 1. / 2, 1. / 2, 1};
   for (int Stage = 0; Stage < 4; ++Stage) {
      const double CB = orc_coeff_seconds(RKB[Stage], dt);
      g_time          = g_sim_time + orc_coeff_seconds(RKC[Stage], dt); /* StageTime :87 */
      if (Stage == 0) {
         orc_weight_tracers(m, NT, s->tr[1], s->tr[0], s->h[0]);
         orc_tend_compute_all(m, c, a, NT, s->hTend, s->uTend, s->trTend, s->h[0], s->u[0], s->tr[0]);
         orc_update_thickness_by_tend(m, s->h[1], s->h[0], s->hTend, CB);
         orc_update_velocity_by_tend(m, s->u[1], s->u[0], s->uTend, CB);
         orc_accumulate_tracers_update(m, NT, s->tr[1], s->trTend, CB);
      } else {
         const double CA = orc_coeff_seconds(RKA[Stage], dt);
         orc_update_thickness_by_tend(m, s->hProvis, s->h[0], s->hTend, CA);
         orc_update_velocity_by_tend(m, s->uProvis, s->u[0], s->uTend, CA);
         orc_update_tracers_by_tend(m, NT, s->trProvis, s->tr[0], s->hProvis, s->h[0], s->trTend, CA);
         if (Stage == 2 && ex)
            ex(ctx, s->hProvis, s->uProvis, s->trProvis);
         orc_tend_compute_all(m, c, a, NT, s->hTend, s->uTend, s->trTend, s->hProvis, s->uProvis, s->trProvis);
         orc_update_thickness_by_tend(m, s->h[1], s->h[1], s->hTend, CB);
         orc_update_velocity_by_tend(m, s->u[1], s->u[1], s->uTend, CB);
         orc_accumulate_tracers_update(m, NT, s->tr[1], s->trTend, CB);
      }
   }
   orc_finalize_tracers_update(m, NT, s->tr[1], s->h[1]);
   if (ex)
      ex(ctx, s->h[1], s->u[1], s->tr[1]); /* State->updateTimeLevels / Tracers::updateTimeLevels :130-131 */
}

/* RungeKutta2Stepper::doStep, O/src/timeStepping/RungeKutta2Stepper.cpp:27-73 */
void orc_rk2_step(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, orc_state *s, double dt, orc_exchange_fn ex, void *ctx) {
   const double CH = orc_coeff_seconds(0.5, dt), C1 = orc_coeff_seconds(1.0, dt);
   g_time = g_sim_time;
   orc_tend_compute_all(m, c, a, NT, s->hTend, s->uTend, s->trTend, s->h[0], s->u[0], s->tr[0]);
   orc_update_thickness_by_tend(m, s->h[1], s->h[0], s->hTend, CH);
   orc_update_velocity_by_tend(m, s->u[1], s->u[0], s->uTend, CH);
   orc_update_tracers_by_tend(m, NT, s->tr[1], s->tr[0], s->h[1], s->h[0], s->trTend, CH);
   g_time = g_sim_time + CH; /* SimTime + 0.5*TimeStep :58 */
   orc_tend_compute_all(m, c, a, NT, s->hTend, s->uTend, s->trTend, s->h[1], s->u[1], s->tr[1]);
   orc_update_thickness_by_tend(m, s->h[1], s->h[0], s->hTend, C1);
   orc_update_velocity_by_tend(m, s->u[1], s->u[0], s->uTend, C1);
   orc_update_tracers_by_tend(m, NT, s->tr[1], s->tr[0], s->h[1], s->h[0], s->trTend, C1);
   if (ex)
      ex(ctx, s->h[1], s->u[1], s->tr[1]);
}

/* ForwardBackwardStepper::doStep, O/src/timeStepping/ForwardBackwardStepper.cpp:27-82 */
void orc_fb_step(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, orc_state *s, double dt, orc_exchange_fn ex, void *ctx) {
   const double C1 = orc_coeff_seconds(1.0, dt);
   g_time = g_sim_time;
   orc_tend_compute_thickness(m, c, a, s->hTend, s->h[0], s->u[0]);
   orc_update_thickness_by_tend(m, s->h[1], s->h[0], s->hTend, C1);
   orc_tend_compute_tracer(m, c, a, NT, s->trTend, s->h[0], s->u[0], s->tr[0]);
   orc_update_tracers_by_tend(m, NT, s->tr[1], s->tr[0], s->h[1], s->h[0], s->trTend, C1);
   g_time = g_sim_time + C1; /* SimTime + TimeStep :67 */
   orc_tend_compute_velocity(m, c, a, s->uTend, s->h[1], s->u[0]);
   orc_update_velocity_by_tend(m, s->u[1], s->u[0], s->uTend, C1);
   if (ex)
      ex(ctx, s->h[1], s->u[1], s->tr[1]);
}
