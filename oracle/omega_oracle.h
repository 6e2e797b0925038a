/* omega_oracle.h -- CPU oracle for the Omega ocean-dycore hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C, double-precision restatement of
 * the reference's functors and launch order; it is imported only by tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg, as the checker.  The
 * product (omega_amd/csrc) never links or calls it.
 *
 * Pinning: the reference cannot be built here (Kokkos, spdlog, SCORPIO, yaml-cpp,
 * cpptrace, METIS are empty/absent), so the oracle is pinned by the
 * reference's own known-answer error norms on the planar 48x48 periodic mesh
 * (test/ocn/TendencyTermsTest.cpp:43-59, AuxiliaryVarsTest.cpp:34-68,
 * HorzOperatorsTest.cpp:33-44) -- see tests/test_oracle_known_answers.py.
 *
 * Every function cites the reference file:line it restates.  "O/" below is
 * /root/reference/components/omega/.  Arrays are LayoutRight (last index = vertical
 * level, contiguous), each mesh-indexed array has NXxSize = NXxAll + 1 rows (the last
 * row is the zero sentinel that missing neighbours point to; O/src/base/Decomp.cpp:553-574).
 * Build with -ffp-contract=off so the operation order below is the rounding order.
 */
#ifndef OMEGA_ORACLE_H
#define OMEGA_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
   /* sizes (O/src/ocn/HorzMesh.h:100-127) */
   int NCellsOwned, NCellsAll, NCellsSize;
   int NEdgesOwned, NEdgesAll, NEdgesSize;
   int NVerticesOwned, NVerticesAll, NVerticesSize;
   int MaxEdges, MaxEdges2, VertexDegree, NVertLayers;
   /* connectivity (O/src/ocn/HorzMesh.h:129-159) */
   const int *NEdgesOnCell;   /* [NCellsSize]                 */
   const int *EdgesOnCell;    /* [NCellsSize][MaxEdges]       */
   const int *CellsOnCell;    /* [NCellsSize][MaxEdges]       */
   const int *VerticesOnCell; /* [NCellsSize][MaxEdges]       */
   const int *CellsOnEdge;    /* [NEdgesSize][2]              */
   const int *VerticesOnEdge; /* [NEdgesSize][2]              */
   const int *NEdgesOnEdge;   /* [NEdgesSize]                 */
   const int *EdgesOnEdge;    /* [NEdgesSize][MaxEdges2]      */
   const int *CellsOnVertex;  /* [NVerticesSize][VertexDegree]*/
   const int *EdgesOnVertex;  /* [NVerticesSize][VertexDegree]*/
   /* geometry (O/src/ocn/HorzMesh.h:192-245) */
   const double *AreaCell, *AreaTriangle, *KiteAreasOnVertex;
   const double *DcEdge, *DvEdge, *AngleEdge, *WeightsOnEdge;
   const double *FVertex, *BottomDepth;
   /* derived (O/src/ocn/HorzMesh.cpp:527-626); filled by orc_mesh_derive */
   double *EdgeSignOnCell;   /* [NCellsSize][MaxEdges]          */
   double *EdgeSignOnVertex; /* [NVerticesSize][VertexDegree]   */
   double *EdgeMask;         /* [NEdgesSize][NVertLayers]       */
   double *MeshScalingDel2;  /* [NEdgesSize] */
   double *MeshScalingDel4;  /* [NEdgesSize] */
} orc_mesh;

/* Tendencies + AuxiliaryState options (O/configs/Default.yml:25-52,
 * O/src/ocn/Tendencies.cpp:123-213, O/src/ocn/AuxiliaryState.cpp:259-308) */
typedef struct {
   int ThicknessFluxTendencyEnable, PVTendencyEnable, KETendencyEnable,
       SSHTendencyEnable, VelDiffTendencyEnable, VelHyperDiffTendencyEnable,
       WindForcingTendencyEnable, BottomDragTendencyEnable,
       TracerHorzAdvTendencyEnable, TracerDiffTendencyEnable,
       TracerHyperDiffTendencyEnable;
   int FluxThicknessUpwind; /* 0 = Center, 1 = Upwind */
   int FluxTracerUpwind;    /* 0 = Center, 1 = Upwind */
   int WindInterpIsotropic; /* 0 = Anisotropic, 1 = Isotropic */
   double ViscDel2, ViscDel4, DivFactor, EddyDiff2, EddyDiff4, Density0,
       BottomDragCoeff;
} orc_config;

/* all AuxiliaryState arrays (O/src/ocn/AuxiliaryState.h:37-42 and auxiliaryVars/ *.h) */
typedef struct {
   double *KineticEnergyCell, *VelocityDivCell;                      /* C x K */
   double *FluxLayerThickEdge, *MeanLayerThickEdge;                  /* E x K */
   double *SshCell;                                                  /* C x K */
   double *RelVortVertex, *NormRelVortVertex, *NormPlanetVortVertex; /* V x K */
   double *NormRelVortEdge, *NormPlanetVortEdge;                     /* E x K */
   double *Del2Edge, *Del2DivCell, *Del2RelVortVertex;               /* E,C,V x K */
   double *HTracersEdge;    /* NT x E x K */
   double *Del2TracersCell; /* NT x C x K */
   double *NormalStressEdge, *ZonalStressCell, *MeridStressCell; /* E, C, C (1-D) */
} orc_aux;

/* ManufacturedSolution (O/src/ocn/CustomTendencyTerms.h, CustomTendencyTerms.cpp:18-210): the custom
 * thickness / velocity tendencies of the manufactured-solution test case (Bishnu et al. 2024) */
typedef struct {
   double H0, Eta0, Kx, Ky, AngFreq, Grav, ViscDel2, ViscDel4;
   int VelDiffTendencyEnable, VelHyperDiffTendencyEnable;
   const double *XCell, *YCell, *XEdge, *YEdge, *FEdge; /* HorzMesh members the functors read (local order) */
} orc_manufactured;
/* fills the constants; the caller sets the five coordinate pointers */
void orc_manufactured_init(orc_manufactured *ms, const orc_mesh *m, const orc_config *c, double WavelengthX,
                           double WavelengthY, double Amplitude);
void orc_manufactured_thickness_tend(const orc_mesh *m, const orc_manufactured *ms, double *hTend, double ElapsedSec);
void orc_manufactured_velocity_tend(const orc_mesh *m, const orc_manufactured *ms, double *uTend, double ElapsedSec);
/* Tendencies::CustomThicknessTend / CustomVelocityTend (Tendencies.cpp:288-291, 416-419): when set, the
 * two group functions add the manufactured terms at the time given by orc_set_time; the steppers set
 * that time per stage from orc_set_sim_time (RungeKutta4Stepper.cpp:87, RungeKutta2Stepper.cpp:44,58,
 * ForwardBackwardStepper.cpp:50,59,67). */
void orc_set_custom_tendency(const orc_manufactured *ms);
/* DecayVelocityTendency (O/test/timeStepping/TimeStepperTest.cpp:49-73) as the custom velocity tendency */
void orc_set_decay_velocity_tendency(int On, double Coeff);
void orc_set_time(double ElapsedSec);
void orc_set_sim_time(double SimTimeSec);

void orc_set_num_threads(int n);
int orc_get_max_threads(void);

void orc_config_default(orc_config *c);
void orc_mesh_derive(orc_mesh *m);

/* ---- HorzOperators (O/src/ocn/HorzOperators.h) on elements [0,N) ---- */
void orc_divergence_on_cell(const orc_mesh *m, int N, double *DivCell, const double *VecEdge);
void orc_gradient_on_edge(const orc_mesh *m, int N, double *GradEdge, const double *ScalarCell);
void orc_curl_on_vertex(const orc_mesh *m, int N, double *CurlVertex, const double *VecEdge);
void orc_tangential_recon_on_edge(const orc_mesh *m, int N, double *ReconEdge, const double *VecEdge);
void orc_interp_cell_to_edge(const orc_mesh *m, int N, double *OutEdge, const double *ArrayCell, int Isotropic);

/* ---- auxiliary variables (O/src/ocn/auxiliaryVars/ *.h) on elements [0,N) ---- */
void orc_vorticity_on_vertex(const orc_mesh *m, int N, const orc_aux *a, const double *h, const double *u);
void orc_vorticity_on_edge(const orc_mesh *m, int N, const orc_aux *a);
void orc_kinetic_on_cell(const orc_mesh *m, int N, const orc_aux *a, const double *u);
void orc_layerthick_on_edge(const orc_mesh *m, int N, const orc_aux *a, const double *h, const double *u, int Upwind);
void orc_layerthick_on_cell(const orc_mesh *m, int N, const orc_aux *a, const double *h);
void orc_veldel2_on_edge(const orc_mesh *m, int N, const orc_aux *a, const double *DivCell, const double *RelVortVertex);
void orc_veldel2_on_cell(const orc_mesh *m, int N, const orc_aux *a);
void orc_veldel2_on_vertex(const orc_mesh *m, int N, const orc_aux *a);
void orc_tracer_on_edge(const orc_mesh *m, int NT, int N, const orc_aux *a, const double *u, const double *h, const double *tr, int Upwind);
void orc_tracer_on_cell(const orc_mesh *m, int NT, int N, const orc_aux *a, const double *hMeanEdge, const double *tr);
void orc_wind_on_edge(const orc_mesh *m, int N, const orc_aux *a, int Isotropic);

/* AuxiliaryState::computeMomAux / computeAll (O/src/ocn/AuxiliaryState.cpp:60-185) */
void orc_aux_compute_mom_aux(const orc_mesh *m, const orc_config *c, const orc_aux *a, const double *h, const double *u);
void orc_aux_compute_all(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, const double *h, const double *u, const double *tr);

/* ---- tendency terms (O/src/ocn/TendencyTerms.h) on elements [0,N); Tend has `TendRows` rows ---- */
void orc_thickness_flux_div_on_cell(const orc_mesh *m, int N, double *Tend, const double *ThicknessFlux, const double *NormalVelEdge);
void orc_pv_hadv_on_edge(const orc_mesh *m, int N, double *Tend, const double *NormRVortEdge, const double *NormFEdge, const double *FluxLayerThickEdge, const double *NormVelEdge);
void orc_ke_grad_on_edge(const orc_mesh *m, int N, double *Tend, const double *KECell);
void orc_ssh_grad_on_edge(const orc_mesh *m, int N, double *Tend, const double *SshCell);
void orc_velocity_diffusion_on_edge(const orc_mesh *m, int N, double *Tend, const double *DivCell, const double *RVortVertex, double ViscDel2);
void orc_velocity_hyperdiff_on_edge(const orc_mesh *m, int N, double *Tend, const double *Del2DivCell, const double *Del2RVortVertex, double ViscDel4, double DivFactor);
void orc_wind_forcing_on_edge(const orc_mesh *m, int N, double *Tend, const double *NormalStressEdge, const double *LayerThickEdge, double SaltWaterDensity);
void orc_bottom_drag_on_edge(const orc_mesh *m, int N, double *Tend, const double *NormalVelEdge, const double *KECell, const double *LayerThickEdge, double Coeff);
/* tracer terms: Tend is [NT][TendRows][K], inputs [NT][InRows][K] */
void orc_tracer_horz_adv_on_cell(const orc_mesh *m, int NT, int N, double *Tend, int TendRows, const double *NormVelEdge, const double *HTracersOnEdge);
void orc_tracer_diff_on_cell(const orc_mesh *m, int NT, int N, double *Tend, int TendRows, const double *TracerCell, const double *MeanLayerThickEdge, double EddyDiff2);
void orc_tracer_hyperdiff_on_cell(const orc_mesh *m, int NT, int N, double *Tend, int TendRows, const double *TrDel2Cell, double EddyDiff4);

/* Tendencies::compute*TendenciesOnly / compute*Tendencies / computeAllTendencies
 * (O/src/ocn/Tendencies.cpp:257-600) */
void orc_tend_thickness_only(const orc_mesh *m, const orc_config *c, const orc_aux *a, double *hTend, const double *u);
void orc_tend_velocity_only(const orc_mesh *m, const orc_config *c, const orc_aux *a, double *uTend, const double *u);
void orc_tend_tracer_only(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, double *trTend, const double *u, const double *tr);
void orc_tend_compute_thickness(const orc_mesh *m, const orc_config *c, const orc_aux *a, double *hTend, const double *h, const double *u);
void orc_tend_compute_velocity(const orc_mesh *m, const orc_config *c, const orc_aux *a, double *uTend, const double *h, const double *u);
void orc_tend_compute_tracer(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, double *trTend, const double *h, const double *u, const double *tr);
void orc_tend_compute_all(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT,
                          double *hTend, double *uTend, double *trTend,
                          const double *h, const double *u, const double *tr);

/* ---- TimeStepper update kernels (O/src/timeStepping/TimeStepper.cpp:378-524) ---- */
void orc_update_thickness_by_tend(const orc_mesh *m, double *h1, const double *h2, const double *hTend, double Coeff);
void orc_update_velocity_by_tend(const orc_mesh *m, double *u1, const double *u2, const double *uTend, double Coeff);
void orc_update_tracers_by_tend(const orc_mesh *m, int NT, double *NextTr, const double *CurTr, const double *h1, const double *h2, const double *trTend, double Coeff);
void orc_weight_tracers(const orc_mesh *m, int NT, double *NextTr, const double *CurTr, const double *hCur);
void orc_accumulate_tracers_update(const orc_mesh *m, int NT, double *AccumTr, const double *trTend, double Coeff);
void orc_finalize_tracers_update(const orc_mesh *m, int NT, double *NextTr, const double *hNext);

/* TimeInterval coefficient: seconds of (Mult * TimeStep) through the reference's
 * integer-fraction arithmetic (O/src/infra/TimeMgr.cpp:193-283, 747-767, 956-1000, 382-391) */
double orc_coeff_seconds(double Mult, double TimeStepSeconds);

/* halo hook for the steppers: called as Exchange(ctx, h, u, tr) where the reference
 * exchanges halos; NULL = single rank, nothing to exchange */
typedef void (*orc_exchange_fn)(void *ctx, double *h, double *u, double *tr);

typedef struct {
   double *h[2], *u[2], *tr[2];  /* two time levels; index 0 = current, 1 = next (CurTimeIndex==0) */
   double *hProvis, *uProvis, *trProvis;
   double *hTend, *uTend, *trTend;
} orc_state;

/* RungeKutta4Stepper::doStep (O/src/timeStepping/RungeKutta4Stepper.cpp:68-137) up to, and including,
 * the end-of-step halo exchange; the caller swaps time levels (OceanState::updateTimeLevels). */
void orc_rk4_step(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, orc_state *s, double dt, orc_exchange_fn ex, void *ctx);
/* RungeKutta2Stepper::doStep (O/src/timeStepping/RungeKutta2Stepper.cpp:27-73) */
void orc_rk2_step(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, orc_state *s, double dt, orc_exchange_fn ex, void *ctx);
/* ForwardBackwardStepper::doStep (O/src/timeStepping/ForwardBackwardStepper.cpp:27-82) */
void orc_fb_step(const orc_mesh *m, const orc_config *c, const orc_aux *a, int NT, orc_state *s, double dt, orc_exchange_fn ex, void *ctx);

#ifdef __cplusplus
}
#endif
#endif
