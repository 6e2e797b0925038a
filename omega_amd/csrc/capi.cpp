// capi.cpp -- extern "C" boundary of libomega_amd.so (see include/omega_amd.h).
#include "../../include/omega_amd.h"

#include "AuxiliaryState.h"
#include "Decomp.h"
#include "Halo.h"
#include "HorzMesh.h"
#include "HorzOperators.h"
#include "OceanState.h"
#include "Tendencies.h"
#include "CustomTendencyTerms.h"
#include "MeshIO.h"
#include "Partition.h"
#include "History.h"
#include "Pacer.h"
#include "PeerWire.h"
#include "Rccl.h"
#include "Tuning.h"
#include "TimeStepper.h"

#include <cstring>
#include <map>
#include <memory>

using namespace OMEGA;

struct omg_decomp {
   std::unique_ptr<Decomp> D;
};
struct omg_halo {
   std::unique_ptr<Halo> H;
};
struct omg_rccl {
   std::unique_ptr<RcclComm> C;
};
struct omg_peer {
   std::unique_ptr<PeerWire> P;
};
struct omg_mesh {
   std::unique_ptr<HorzMesh> M;
};
struct omg_state {
   std::unique_ptr<OceanState> S;
};
struct omg_tracers {
   std::unique_ptr<TracerStore> T;
};
struct omg_aux {
   std::unique_ptr<AuxiliaryState> A;
};
struct omg_mesh_file {
   std::unique_ptr<MeshFile> F;
};
struct omg_tend {
   std::unique_ptr<Tendencies> T;
   std::shared_ptr<ManufacturedSolution> Ms;
};
struct omg_stepper {
   std::unique_ptr<TimeStepper> St;
};

static thread_local std::string LastError;

#define OMG_TRY try {
#define OMG_CATCH                                                              \
   }                                                                           \
   catch (const std::exception &E) {                                           \
      LastError = E.what();                                                    \
      return 1;                                                                \
   }                                                                           \
   catch (...) {                                                               \
      LastError = "unknown exception";                                         \
      return 1;                                                                \
   }                                                                           \
   return 0;

#define OMG_ARG(cond)                                                          \
   if (!(cond))                                                                \
   OMEGA_ABORT(std::string("invalid argument: ") + #cond)

/// a device array as (pointer, rows of the last index, width, row pitch) -- see copyRowsToHost
struct ArrRef {
   Real *Ptr;
   size_t Rows;
   int Width, Pitch;
   size_t size() const { return Rows * (size_t)Width; }
};
template <int N> static ArrRef arrRef(const DeviceArray<Real, N> &A) {
   return ArrRef{A.Ptr, A.rows(), A.Ext[N - 1], A.Pitch};
}

extern "C" {

const char *omg_last_error(void) { return LastError.c_str(); }

int omg_device_count(int *n) {
   OMG_TRY
   OMG_ARG(n);
   int C = 0;
   if (hipGetDeviceCount(&C) != hipSuccess)
      C = 0;
   *n = C;
   OMG_CATCH
}
int omg_device_init(int device_id) {
   OMG_TRY
   deviceInit(device_id);
   OMG_CATCH
}
int omg_device_synchronize(void) {
   OMG_TRY
   HIP_CHECK(hipDeviceSynchronize());
   OMG_CATCH
}
int omg_stream_create(void **stream) {
   OMG_TRY
   OMG_ARG(stream);
   hipStream_t S;
   HIP_CHECK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
   *stream = (void *)S;
   OMG_CATCH
}
int omg_stream_destroy(void *stream) {
   OMG_TRY
   if (stream)
      HIP_CHECK(hipStreamDestroy((hipStream_t)stream));
   OMG_CATCH
}
int omg_stream_synchronize(void *stream) {
   OMG_TRY
   HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
   OMG_CATCH
}
int omg_event_create(void **event) {
   OMG_TRY
   OMG_ARG(event);
   hipEvent_t E;
   HIP_CHECK(hipEventCreate(&E));
   *event = (void *)E;
   OMG_CATCH
}
int omg_event_destroy(void *event) {
   OMG_TRY
   if (event)
      HIP_CHECK(hipEventDestroy((hipEvent_t)event));
   OMG_CATCH
}
int omg_event_record(void *event, void *stream) {
   OMG_TRY
   HIP_CHECK(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
   OMG_CATCH
}
int omg_event_elapsed_ms(void *start, void *stop, float *ms) {
   OMG_TRY
   OMG_ARG(ms);
   HIP_CHECK(hipEventSynchronize((hipEvent_t)stop));
   HIP_CHECK(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
   OMG_CATCH
}

// ---------------------------------------------------------------- raw device buffers
int omg_device_resource_count(int64_t *n) {
   OMG_TRY
   OMG_ARG(n);
   *n = deviceResourceCount();
   OMG_CATCH
}
int omg_device_malloc(size_t bytes, void **ptr) {
   OMG_TRY
   OMG_ARG(ptr);
   HIP_CHECK(hipMalloc(ptr, bytes ? bytes : 1));
   OMG_CATCH
}
int omg_device_free(void *ptr) {
   OMG_TRY
   if (ptr)
      HIP_CHECK(hipFree(ptr));
   OMG_CATCH
}
int omg_copy_to_device(void *dst, const void *src, size_t bytes) {
   OMG_TRY
   OMG_ARG(dst && src);
   HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice));
   OMG_CATCH
}
int omg_copy_to_host(void *dst, const void *src, size_t bytes) {
   OMG_TRY
   OMG_ARG(dst && src);
   HIP_CHECK(hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost));
   OMG_CATCH
}

// ---------------------------------------------------------------- reductions
int omg_local_sum_dd(const double *a, const double *b, size_t n, void *stream, double *hi_lo) {
   OMG_TRY
   OMG_ARG(a && hi_lo);
   localSumDD(a, b, n, (hipStream_t)stream, hi_lo);
   OMG_CATCH
}
int omg_local_weighted_sum_dd(const double *w, const double *a, const double *b, int nrows, int k, int row_pitch,
                              void *stream, double *hi_lo) {
   OMG_TRY
   OMG_ARG(w && a && hi_lo && nrows >= 0 && k > 0 && (row_pitch == 0 || row_pitch >= k));
   localWeightedSumDD(w, a, b, nrows, k, row_pitch > 0 ? row_pitch : k, (hipStream_t)stream, hi_lo);
   OMG_CATCH
}
int omg_combine_dd(const double *pairs, int npairs, double *hi_lo) {
   OMG_TRY
   OMG_ARG(pairs && hi_lo && npairs >= 0);
   combineDD(pairs, npairs, hi_lo);
   OMG_CATCH
}

// ---------------------------------------------------------------- mesh file
int omg_mesh_file_open(const char *path, omg_mesh_file **out) {
   OMG_TRY
   OMG_ARG(path && out);
   auto *R = new omg_mesh_file;
   try {
      R->F.reset(new MeshFile(path));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_mesh_file_close(omg_mesh_file *f) {
   delete f;
   return 0;
}
int omg_mesh_file_global_mesh(const omg_mesh_file *f, omg_global_mesh *m) {
   OMG_TRY
   OMG_ARG(f && m);
   const GlobalMeshDesc &G = f->F->desc();
   m->nCells = G.NCells, m->nEdges = G.NEdges, m->nVertices = G.NVertices, m->maxEdges = G.MaxEdges;
   m->vertexDegree = G.VertexDegree;
   m->cellsOnCell = G.CellsOnCell, m->edgesOnCell = G.EdgesOnCell, m->verticesOnCell = G.VerticesOnCell;
   m->cellsOnEdge = G.CellsOnEdge, m->verticesOnEdge = G.VerticesOnEdge, m->edgesOnEdge = G.EdgesOnEdge;
   m->cellsOnVertex = G.CellsOnVertex, m->edgesOnVertex = G.EdgesOnVertex;
   m->xCell = G.XCell, m->yCell = G.YCell, m->zCell = G.ZCell, m->lonCell = G.LonCell, m->latCell = G.LatCell;
   m->xEdge = G.XEdge, m->yEdge = G.YEdge, m->zEdge = G.ZEdge, m->lonEdge = G.LonEdge, m->latEdge = G.LatEdge;
   m->xVertex = G.XVertex, m->yVertex = G.YVertex, m->zVertex = G.ZVertex, m->lonVertex = G.LonVertex;
   m->latVertex = G.LatVertex;
   m->areaCell = G.AreaCell, m->areaTriangle = G.AreaTriangle, m->kiteAreasOnVertex = G.KiteAreasOnVertex;
   m->dcEdge = G.DcEdge, m->dvEdge = G.DvEdge, m->angleEdge = G.AngleEdge, m->weightsOnEdge = G.WeightsOnEdge;
   m->fCell = G.FCell, m->fEdge = G.FEdge, m->fVertex = G.FVertex, m->bottomDepth = G.BottomDepth;
   OMG_CATCH
}
int omg_mesh_file_dim(const omg_mesh_file *f, const char *name, int64_t *len) {
   OMG_TRY
   OMG_ARG(f && name && len);
   *len = f->F->file().hasDim(name) ? f->F->file().dimLen(name) : -1;
   OMG_CATCH
}
int omg_mesh_file_var_size(const omg_mesh_file *f, const char *name, int64_t record, int64_t *n) {
   OMG_TRY
   OMG_ARG(f && name && n);
   if (!f->F->file().hasVar(name)) {
      *n = -1;
   } else {
      const std::vector<I8> S = f->F->file().shape(name);
      const bool Rec          = f->F->file().var(name).IsRecord;
      I8 N                    = 1;
      for (size_t D = (Rec && record >= 0) ? 1 : 0; D < S.size(); ++D)
         N *= S[D];
      *n = N;
   }
   OMG_CATCH
}
int omg_mesh_file_read_f64(const omg_mesh_file *f, const char *name, int64_t record, double *out, size_t n) {
   OMG_TRY
   OMG_ARG(f && name && out);
   std::vector<R8> V;
   f->F->file().read(name, V, record);
   if (V.size() != n)
      OMEGA_ABORT(std::string("omg_mesh_file_read_f64: ") + name + " has " + std::to_string(V.size()) +
                  " values, buffer holds " + std::to_string(n));
   std::memcpy(out, V.data(), n * sizeof(double));
   OMG_CATCH
}

// ---------------------------------------------------------------- restart file
struct omg_restart_file {
   std::unique_ptr<RestartFile> F;
};
int omg_restart_create(const char *path, int64_t ncells_global, int64_t nedges_global, int nvertlevels, int ntracers,
                       double simulation_time, int64_t steps_done) {
   OMG_TRY
   OMG_ARG(path);
   RestartFile::create(path, ncells_global, nedges_global, nvertlevels, ntracers, simulation_time, steps_done);
   OMG_CATCH
}
int omg_restart_open(const char *path, int write, omg_restart_file **out) {
   OMG_TRY
   OMG_ARG(path && out);
   auto *R = new omg_restart_file;
   try {
      R->F.reset(new RestartFile(path, write != 0));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_restart_close(omg_restart_file *f) {
   delete f;
   return 0;
}
int omg_restart_info(const omg_restart_file *f, int64_t *ncells, int64_t *nedges, int *nvertlevels, int *ntracers,
                     double *simulation_time, int64_t *steps_done) {
   OMG_TRY
   OMG_ARG(f && ncells && nedges && nvertlevels && ntracers && simulation_time && steps_done);
   *ncells = f->F->NCells, *nedges = f->F->NEdges, *nvertlevels = f->F->NVertLevels, *ntracers = f->F->NTracers;
   *simulation_time = f->F->SimulationTime, *steps_done = f->F->StepsDone;
   OMG_CATCH
}
int omg_restart_write_rows(omg_restart_file *f, const char *var, int plane, const int32_t *global_id, int64_t n,
                           const double *rows) {
   OMG_TRY
   OMG_ARG(f && var && global_id && rows);
   f->F->writeRows(var, plane, global_id, n, rows);
   OMG_CATCH
}
int omg_restart_read_rows(const omg_restart_file *f, const char *var, int plane, const int32_t *global_id, int64_t n,
                          double *rows) {
   OMG_TRY
   OMG_ARG(f && var && global_id && rows);
   f->F->readRows(var, plane, global_id, n, rows);
   OMG_CATCH
}

// ---------------------------------------------------------------- Decomp
static GlobalMeshDesc toDesc(const omg_global_mesh &M) {
   const omg_global_mesh *m = &M;
   GlobalMeshDesc G;
   G.NCells = m->nCells, G.NEdges = m->nEdges, G.NVertices = m->nVertices, G.MaxEdges = m->maxEdges;
   G.VertexDegree = m->vertexDegree;
   G.CellsOnCell = m->cellsOnCell, G.EdgesOnCell = m->edgesOnCell, G.VerticesOnCell = m->verticesOnCell;
   G.CellsOnEdge = m->cellsOnEdge, G.VerticesOnEdge = m->verticesOnEdge, G.EdgesOnEdge = m->edgesOnEdge;
   G.CellsOnVertex = m->cellsOnVertex, G.EdgesOnVertex = m->edgesOnVertex;
   G.XCell = m->xCell, G.YCell = m->yCell, G.ZCell = m->zCell, G.LonCell = m->lonCell, G.LatCell = m->latCell;
   G.XEdge = m->xEdge, G.YEdge = m->yEdge, G.ZEdge = m->zEdge, G.LonEdge = m->lonEdge, G.LatEdge = m->latEdge;
   G.XVertex = m->xVertex, G.YVertex = m->yVertex, G.ZVertex = m->zVertex, G.LonVertex = m->lonVertex;
   G.LatVertex = m->latVertex;
   G.AreaCell = m->areaCell, G.AreaTriangle = m->areaTriangle, G.KiteAreasOnVertex = m->kiteAreasOnVertex;
   G.DcEdge = m->dcEdge, G.DvEdge = m->dvEdge, G.AngleEdge = m->angleEdge, G.WeightsOnEdge = m->weightsOnEdge;
   G.FCell = m->fCell, G.FEdge = m->fEdge, G.FVertex = m->fVertex, G.BottomDepth = m->bottomDepth;
   return G;
}
int omg_decomp_create_ordered(const omg_global_mesh *mesh, int nparts, int mytask, int halo_width,
                              const int32_t *cell_task, int local_order, omg_decomp **out);
int omg_decomp_create(const omg_global_mesh *m, int nparts, int mytask, int halo_width, const int32_t *cell_task,
                      omg_decomp **out) {
   return omg_decomp_create_ordered(m, nparts, mytask, halo_width, cell_task, 0, out);
}
int omg_decomp_create_ordered(const omg_global_mesh *mesh, int nparts, int mytask, int halo_width,
                              const int32_t *cell_task, int local_order, omg_decomp **out) {
   OMG_TRY
   OMG_ARG(mesh && out && local_order >= 0 && local_order <= 3);
   auto *R = new omg_decomp;
   try {
      R->D.reset(new Decomp(toDesc(*mesh), nparts, mytask, halo_width, cell_task, (LocalOrder)local_order));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_partition_cells(const omg_global_mesh *mesh, int nparts, const char *method, int32_t *cell_task_out,
                        int64_t *edge_cut) {
   OMG_TRY
   OMG_ARG(mesh && method && cell_task_out && nparts >= 1);
   const GlobalMeshDesc G = toDesc(*mesh);
   std::vector<I4> T;
   const std::string M(method);
   if (M == "graph") {
      partitionGraph(G, nparts, T);
   } else if (M == "rcb") {
      partitionRCB(G, nparts, T);
   } else {
      OMEGA_ABORT("omg_partition_cells: method must be \"rcb\" or \"graph\"");
   }
   std::memcpy(cell_task_out, T.data(), T.size() * sizeof(I4));
   if (edge_cut)
      *edge_cut = edgeCut(G, T);
   OMG_CATCH
}
int omg_decomp_destroy(omg_decomp *d) {
   delete d;
   return 0;
}
int omg_decomp_get_int(const omg_decomp *d, const char *name, int32_t *out) {
   OMG_TRY
   OMG_ARG(d && name && out);
   const Decomp &D = *d->D;
   const std::map<std::string, I4> V{{"NCellsGlobal", D.NCellsGlobal},
                                     {"NCellsOwned", D.NCellsOwned},
                                     {"NCellsAll", D.NCellsAll},
                                     {"NCellsSize", D.NCellsSize},
                                     {"NEdgesGlobal", D.NEdgesGlobal},
                                     {"NEdgesOwned", D.NEdgesOwned},
                                     {"NEdgesAll", D.NEdgesAll},
                                     {"NEdgesSize", D.NEdgesSize},
                                     {"NVerticesGlobal", D.NVerticesGlobal},
                                     {"NVerticesOwned", D.NVerticesOwned},
                                     {"NVerticesAll", D.NVerticesAll},
                                     {"NVerticesSize", D.NVerticesSize},
                                     {"MaxEdges", D.MaxEdges},
                                     {"VertexDegree", D.VertexDegree},
                                     {"HaloWidth", D.HaloWidth},
                                     {"NumTasks", D.NumTasks},
                                     {"MyTask", D.MyTask}};
   auto It = V.find(name);
   if (It == V.end())
      OMEGA_ABORT(std::string("Decomp: no integer member named ") + name);
   *out = It->second;
   OMG_CATCH
}
static void copyOutI4(const I4 *Src, size_t Cnt, int32_t *Out, size_t N, const char *Name) {
   if (N < Cnt)
      OMEGA_ABORT(std::string("output buffer too small for ") + Name);
   std::memcpy(Out, Src, Cnt * sizeof(I4));
}
int omg_decomp_get_array(const omg_decomp *d, const char *name, int32_t *out, size_t n) {
   OMG_TRY
   OMG_ARG(d && name && out);
   const Decomp &D = *d->D;
   const std::string S(name);
   const std::map<std::string, const HostArrayI4 *> A{
       {"CellID", &D.CellIDH},           {"EdgeID", &D.EdgeIDH},         {"VertexID", &D.VertexIDH},
       {"CellLoc", &D.CellLocH},         {"EdgeLoc", &D.EdgeLocH},       {"VertexLoc", &D.VertexLocH},
       {"NCellsHalo", &D.NCellsHaloH},   {"NEdgesHalo", &D.NEdgesHaloH}, {"NVerticesHalo", &D.NVerticesHaloH}};
   auto It = A.find(S);
   if (It != A.end())
      copyOutI4(It->second->data(), It->second->size(), out, n, name);
   else if (S == "CellTask")
      copyOutI4(D.CellTask.data(), D.CellTask.size(), out, n, name);
   else
      OMEGA_ABORT("Decomp: no array member named " + S);
   OMG_CATCH
}

// ---------------------------------------------------------------- Halo
int omg_halo_create(const omg_decomp *d, omg_halo **out) {
   OMG_TRY
   OMG_ARG(d && out);
   auto *R = new omg_halo;
   try {
      R->H.reset(new Halo("Default", d->D.get()));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_halo_destroy(omg_halo *h) {
   delete h;
   return 0;
}
int omg_halo_num_neighbors(const omg_halo *h, int *n) {
   OMG_TRY
   OMG_ARG(h && n);
   *n = h->H->NNghbr;
   OMG_CATCH
}
int omg_halo_neighbor_task(const omg_halo *h, int i, int *task) {
   OMG_TRY
   OMG_ARG(h && task && i >= 0 && i < h->H->NNghbr);
   *task = h->H->NeighborList[i];
   OMG_CATCH
}
int omg_halo_list_size(const omg_halo *h, int i, int elem, int recv, int *n) {
   OMG_TRY
   OMG_ARG(h && n && i >= 0 && i < h->H->NNghbr && elem >= 0 && elem < 3);
   *n = (int)(recv ? h->H->RecvLists[elem][i] : h->H->SendLists[elem][i]).size();
   OMG_CATCH
}
int omg_halo_get_list(const omg_halo *h, int i, int elem, int recv, int32_t *out) {
   OMG_TRY
   OMG_ARG(h && out && i >= 0 && i < h->H->NNghbr && elem >= 0 && elem < 3);
   const auto &L = recv ? h->H->RecvLists[elem][i] : h->H->SendLists[elem][i];
   std::memcpy(out, L.data(), L.size() * sizeof(I4));
   OMG_CATCH
}
int omg_halo_required_bytes(const omg_halo *h, int i, size_t pc, size_t pe, size_t pv, size_t *bytes) {
   OMG_TRY
   OMG_ARG(h && bytes && i >= 0 && i < h->H->NNghbr);
   *bytes = h->H->requiredBytes(i, pc, pe, pv);
   OMG_CATCH
}
int omg_history_write(const char *path, const omg_decomp *d, const omg_state *s, const omg_tracers *t, omg_aux *a,
                      const char *contents, double simulation_time, int time_level, int create_file, void *stream,
                      int *n_variables_written) {
   OMG_TRY
   OMG_ARG(path && d && s && a && contents);
   const int N = writeHistory(path, d->D.get(), s->S.get(), t ? t->T.get() : nullptr, a->A.get(), contents,
                              simulation_time, time_level, create_file != 0, (hipStream_t)stream);
   if (n_variables_written)
      *n_variables_written = N;
   OMG_CATCH
}
int omg_rccl_get_unique_id(char *id) {
   OMG_TRY
   OMG_ARG(id);
   RcclComm::getUniqueId(id);
   OMG_CATCH
}
int omg_rccl_create(const char *id, int nranks, int rank, omg_rccl **out) {
   OMG_TRY
   OMG_ARG(id && out);
   auto *R = new omg_rccl;
   try {
      R->C.reset(new RcclComm(id, nranks, rank));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_rccl_destroy(omg_rccl *c) {
   delete c;
   return 0;
}
int omg_rccl_abort(omg_rccl *c) {
   OMG_TRY
   OMG_ARG(c);
   c->C->abort();
   OMG_CATCH
}
int omg_rccl_info(const omg_rccl *c, int *nranks, int *rank, int *version, int64_t *exchanges) {
   OMG_TRY
   OMG_ARG(c);
   if (nranks)
      *nranks = c->C->NRanks;
   if (rank)
      *rank = c->C->Rank;
   if (version)
      *version = c->C->Version;
   if (exchanges)
      *exchanges = c->C->NExchanges;
   OMG_CATCH
}
int omg_rccl_exchange(omg_rccl *c, int n, const int *peers, void *const *send_ptrs, const size_t *send_bytes,
                      void *const *recv_ptrs, const size_t *recv_bytes, void *stream) {
   OMG_TRY
   OMG_ARG(c && n >= 0 && (n == 0 || (peers && send_ptrs && send_bytes && recv_ptrs && recv_bytes)));
   if (c->C->exchange(n, peers, send_ptrs, send_bytes, recv_ptrs, recv_bytes, (hipStream_t)stream) != 0)
      OMEGA_ABORT(c->C->lastError());
   OMG_CATCH
}
int omg_halo_use_rccl(omg_halo *h, omg_rccl *c) {
   OMG_TRY
   OMG_ARG(h && c);
   h->H->useRccl(c->C.get());
   OMG_CATCH
}
int omg_set_option(const char *name, int value) {
   OMG_TRY
   OMG_ARG(name);
   if (!setTuningOption(name, value))
      OMEGA_ABORT(std::string("omg_set_option: no option named '") + name + "'");
   OMG_CATCH
}
int omg_get_option(const char *name, int *value) {
   OMG_TRY
   OMG_ARG(name && value);
   if (!getTuningOption(name, *value))
      OMEGA_ABORT(std::string("omg_get_option: no option named '") + name + "'");
   OMG_CATCH
}
int omg_set_timing_level(int level) {
   Pacer::timingLevel() = level;
   return 0;
}
int omg_peer_create(int nranks, int rank, size_t mailbox_bytes, omg_peer **out) {
   OMG_TRY
   OMG_ARG(out);
   auto *R = new omg_peer;
   try {
      R->P.reset(new PeerWire(nranks, rank, mailbox_bytes));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_peer_destroy(omg_peer *p) {
   delete p;
   return 0;
}
int omg_peer_local_handle(const omg_peer *p, char *out) {
   OMG_TRY
   OMG_ARG(p && out);
   p->P->localHandle(out);
   OMG_CATCH
}
int omg_peer_connect(omg_peer *p, const char *all_handles) {
   OMG_TRY
   OMG_ARG(p && all_handles);
   p->P->connect(all_handles);
   OMG_CATCH
}
int omg_peer_info(const omg_peer *p, int64_t *exchanges, int *status) {
   OMG_TRY
   OMG_ARG(p);
   if (exchanges)
      *exchanges = p->P->NExchanges;
   if (status)
      *status = p->P->status();
   OMG_CATCH
}
int omg_peer_set_timeout(omg_peer *p, double seconds) {
   OMG_TRY
   OMG_ARG(p);
   p->P->setTimeout(seconds);
   OMG_CATCH
}
int omg_halo_use_peer(omg_halo *h, omg_peer *p) {
   OMG_TRY
   OMG_ARG(h && p);
   h->H->usePeerWire(p->P.get());
   OMG_CATCH
}
int omg_halo_recv_rows(const omg_halo *h, size_t per_cell, size_t per_edge, size_t per_vertex, size_t *rows) {
   OMG_TRY
   OMG_ARG(h && rows);
   *rows = h->H->recvRows(per_cell, per_edge, per_vertex);
   OMG_CATCH
}
int omg_halo_set_transport(omg_halo *h, omg_transport_fn fn, void *ctx) {
   OMG_TRY
   OMG_ARG(h);
   h->H->setTransport((HaloTransportFn)fn, ctx);
   OMG_CATCH
}
int omg_halo_exchange(omg_halo *h, double *dev_array, int nt, int rows_size, int k, int row_pitch, int elem,
                      void *stream) {
   OMG_TRY
   OMG_ARG(h && dev_array && nt >= 1 && elem >= 0 && elem < 3 && k >= 1 && (row_pitch == 0 || row_pitch >= k));
   if (h->H->exchangeRaw(dev_array, nt, rows_size, k, row_pitch, (MeshElement)elem, (hipStream_t)stream) != 0)
      OMEGA_ABORT("Halo::exchangeFullArrayHalo failed" + h->H->wireError());
   OMG_CATCH
}

int omg_halo_exchange_bytes(omg_halo *h, void *dev_array, int elem_bytes, int nt, int rows_size, int k, int row_pitch,
                            int elem, void *stream) {
   OMG_TRY
   OMG_ARG(h && dev_array && (elem_bytes == 4 || elem_bytes == 8) && nt >= 1 && elem >= 0 && elem < 3 && k >= 1 &&
           (row_pitch == 0 || row_pitch >= k));
   if (h->H->exchangeRawBytes(dev_array, elem_bytes, nt, rows_size, k, row_pitch, (MeshElement)elem, (hipStream_t)stream) != 0)
      OMEGA_ABORT("Halo::exchangeFullArrayHalo failed" + h->H->wireError());
   OMG_CATCH
}
int omg_halo_exchange_i4(omg_halo *h, int32_t *dev_array, int nt, int rows_size, int k, int row_pitch, int elem,
                         void *stream) {
   return omg_halo_exchange_bytes(h, dev_array, 4, nt, rows_size, k, row_pitch, elem, stream);
}
int omg_halo_global_sum_dd(omg_halo *h, const double *local_hi_lo, int npairs, double *hi_lo, void *stream) {
   OMG_TRY
   OMG_ARG(h && local_hi_lo && hi_lo && npairs >= 0);
   if (h->H->globalSumDD(local_hi_lo, npairs, hi_lo, (hipStream_t)stream) != 0)
      OMEGA_ABORT("Halo::globalSumDD failed" + h->H->wireError());
   OMG_CATCH
}
int omg_halo_check(const omg_halo *h) {
   OMG_TRY
   OMG_ARG(h);
   if (h->H->checkWire() != 0)
      OMEGA_ABORT("Halo: the wire reports a failed exchange" + h->H->wireError());
   OMG_CATCH
}

// ---------------------------------------------------------------- HorzMesh
int omg_mesh_create(const omg_decomp *d, int nvertlayers, int host_only, omg_mesh **out) {
   OMG_TRY
   OMG_ARG(d && out);
   auto *R = new omg_mesh;
   try {
      R->M.reset(new HorzMesh("Default", d->D.get(), nvertlayers, host_only != 0));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_mesh_destroy(omg_mesh *m) {
   delete m;
   return 0;
}
int omg_mesh_get_int(const omg_mesh *m, const char *name, int32_t *out) {
   OMG_TRY
   OMG_ARG(m && name && out);
   const HorzMesh &M = *m->M;
   const std::map<std::string, I4> V{{"NCellsOwned", M.NCellsOwned},       {"NCellsAll", M.NCellsAll},
                                     {"NCellsSize", M.NCellsSize},         {"NEdgesOwned", M.NEdgesOwned},
                                     {"NEdgesAll", M.NEdgesAll},           {"NEdgesSize", M.NEdgesSize},
                                     {"NVerticesOwned", M.NVerticesOwned}, {"NVerticesAll", M.NVerticesAll},
                                     {"NVerticesSize", M.NVerticesSize},   {"MaxEdges", M.MaxEdges},
                                     {"MaxEdgesFile", M.MaxEdgesFile},
                                     {"MaxEdges2", M.MaxEdges2},           {"VertexDegree", M.VertexDegree},
                                     {"NVertLayers", M.NVertLayers},       {"MaxCellsOnEdge", M.MaxCellsOnEdge}};
   if (!M.HostOnly) { // kernel-side table statistics (diagnostics)
      const MeshView &W = M.view();
      const std::map<std::string, I4> D{{"PVChainOK", W.PVChainOK},
                                        {"CellPVOK", W.CellPVOK},
                                        {"CellL1OK", W.CellL1OK},
                                        {"CellPVFinalOK", W.CellPVFinalOK},
                                        {"NIrregularEdges", W.NIrregularEdges},
                                        {"NIrregularOwned", W.NIrregularOwned},
                                        {"NIrregularInner", W.NIrregularInner},
                                        {"DomM1", W.DomM1},
                                        {"NWideCells", W.NWideCells},
                                        {"NBadCells", W.NBadCells},
                                        {"NOrphanVertices", W.NOrphanVertices},
                                        {"NarrowTables", M.narrowView() ? 1 : 0},
                                        {"Del2RingOK", W.Del2RingOK},
                                        {"Del2VertOK", W.Del2VertOK},
                                        {"NBandCells", W.NBandCells},
                                        {"NBandSendCells", W.NBandSendCells},
                                        {"NInteriorCells", W.NInteriorCells},
                                        {"NPatchTiles8", M.NPatchTiles[0]},
                                        {"NPatchTiles16", M.NPatchTiles[1]},
                                        {"NPatchTiles32", M.NPatchTiles[2]},
                                        {"NPatchFallback8", M.NPatchFallback[0]},
                                        {"NPatchFallback16", M.NPatchFallback[1]},
                                        {"NPatchFallback32", M.NPatchFallback[2]}};
      auto Jt = D.find(name);
      if (Jt != D.end()) {
         *out = Jt->second;
         return 0;
      }
   }
   auto It = V.find(name);
   if (It == V.end())
      OMEGA_ABORT(std::string("HorzMesh: no integer member named ") + name);
   *out = It->second;
   OMG_CATCH
}
int omg_mesh_get_array_i4(const omg_mesh *m, const char *name, int32_t *out, size_t n) {
   OMG_TRY
   OMG_ARG(m && name && out);
   const HorzMesh &M = *m->M;
   const std::map<std::string, const HostArrayI4 *> A{
       {"CellsOnCell", &M.CellsOnCellH},       {"EdgesOnCell", &M.EdgesOnCellH},   {"NEdgesOnCell", &M.NEdgesOnCellH},
       {"VerticesOnCell", &M.VerticesOnCellH}, {"CellsOnEdge", &M.CellsOnEdgeH},   {"EdgesOnEdge", &M.EdgesOnEdgeH},
       {"NEdgesOnEdge", &M.NEdgesOnEdgeH},     {"VerticesOnEdge", &M.VerticesOnEdgeH},
       {"CellsOnVertex", &M.CellsOnVertexH},   {"EdgesOnVertex", &M.EdgesOnVertexH},
       {"NCellsHalo", &M.NCellsHaloH},         {"NEdgesHalo", &M.NEdgesHaloH},     {"NVerticesHalo", &M.NVerticesHaloH}};
   auto It = A.find(name);
   if (It == A.end())
      OMEGA_ABORT(std::string("HorzMesh: no int array member named ") + name);
   copyOutI4(It->second->data(), It->second->size(), out, n, name);
   OMG_CATCH
}
int omg_mesh_get_array_r8(const omg_mesh *m, const char *name, double *out, size_t n) {
   OMG_TRY
   OMG_ARG(m && name && out);
   const HorzMesh &M = *m->M;
   const std::map<std::string, const HostArrayReal *> A{
       {"XCell", &M.XCellH},           {"YCell", &M.YCellH},       {"ZCell", &M.ZCellH},
       {"LonCell", &M.LonCellH},       {"LatCell", &M.LatCellH},   {"XEdge", &M.XEdgeH},
       {"YEdge", &M.YEdgeH},           {"ZEdge", &M.ZEdgeH},       {"LonEdge", &M.LonEdgeH},
       {"LatEdge", &M.LatEdgeH},       {"XVertex", &M.XVertexH},   {"YVertex", &M.YVertexH},
       {"ZVertex", &M.ZVertexH},       {"LonVertex", &M.LonVertexH}, {"LatVertex", &M.LatVertexH},
       {"AreaCell", &M.AreaCellH},     {"AreaTriangle", &M.AreaTriangleH},
       {"KiteAreasOnVertex", &M.KiteAreasOnVertexH},               {"DvEdge", &M.DvEdgeH},
       {"DcEdge", &M.DcEdgeH},         {"AngleEdge", &M.AngleEdgeH}, {"WeightsOnEdge", &M.WeightsOnEdgeH},
       {"FEdge", &M.FEdgeH},           {"FCell", &M.FCellH},       {"FVertex", &M.FVertexH},
       {"BottomDepth", &M.BottomDepthH}, {"EdgeSignOnCell", &M.EdgeSignOnCellH},
       {"EdgeSignOnVertex", &M.EdgeSignOnVertexH},                 {"EdgeMask1D", &M.EdgeMask1DH},
       {"MeshScalingDel2", &M.MeshScalingDel2H},                   {"MeshScalingDel4", &M.MeshScalingDel4H}};
   if (std::string(name) == "EdgeMask") { // expanded to the reference's (NEdgesSize, NVertLayers) shape
      const HostArrayReal M2 = M.edgeMask2D();
      if (n < M2.size())
         OMEGA_ABORT("output buffer too small for EdgeMask");
      std::memcpy(out, M2.data(), M2.size() * sizeof(double));
      return 0;
   }
   auto It = A.find(name);
   if (It == A.end())
      OMEGA_ABORT(std::string("HorzMesh: no real array member named ") + name);
   if (n < It->second->size())
      OMEGA_ABORT(std::string("output buffer too small for ") + name);
   std::memcpy(out, It->second->data(), It->second->size() * sizeof(double));
   OMG_CATCH
}
int omg_mesh_set_fvertex(omg_mesh *m, const double *host_values) {
   OMG_TRY
   OMG_ARG(m && host_values);
   m->M->setFVertex(host_values);
   OMG_CATCH
}

static void requireDevice(const HorzMesh *M) {
   OMEGA_REQUIRE(!M->HostOnly, "this mesh was created host-only: no device arrays, compute is unavailable");
}

// ---------------------------------------------------------------- options
#define OMG_HORZ_OP(NAME, LAUNCH, NALL)                                                                              \
   int NAME(const omg_mesh *m, const double *in, double *out, int k, int row_pitch, int n, void *stream) {         \
      OMG_TRY                                                                                                      \
      OMG_ARG(m && in && out && k > 0 && (row_pitch == 0 || row_pitch >= k));                                      \
      OMEGA_REQUIRE(!m->M->HostOnly, #NAME ": needs a device mesh");                                               \
      OMG_ARG(n <= m->M->NALL);                                                                                    \
      LAUNCH(m->M->view(), n < 0 ? m->M->NALL : n, k, row_pitch > 0 ? row_pitch : k, out, in, (hipStream_t)stream); \
      OMG_CATCH                                                                                                    \
   }
OMG_HORZ_OP(omg_horz_divergence, launchDivergenceOnCell, NCellsAll)
OMG_HORZ_OP(omg_horz_gradient, launchGradientOnEdge, NEdgesAll)
OMG_HORZ_OP(omg_horz_curl, launchCurlOnVertex, NVerticesAll)
OMG_HORZ_OP(omg_horz_tangential_recon, launchTangentialReconOnEdge, NEdgesAll)
#undef OMG_HORZ_OP
int omg_horz_interp_cell_to_edge(const omg_mesh *m, const double *cell, double *edge, int isotropic, int n, void *stream) {
   OMG_TRY
   OMG_ARG(m && cell && edge && n <= m->M->NEdgesAll);
   OMEGA_REQUIRE(!m->M->HostOnly, "omg_horz_interp_cell_to_edge: needs a device mesh");
   launchInterpCellToEdge(m->M->view(), n < 0 ? m->M->NEdgesAll : n, edge, cell, isotropic, (hipStream_t)stream);
   OMG_CATCH
}

void omg_tend_config_default(omg_tend_config *c) {
   const TendParams P;
   c->ThicknessFluxTendencyEnable   = P.ThicknessFluxTendencyEnable;
   c->PVTendencyEnable              = P.PVTendencyEnable;
   c->KETendencyEnable              = P.KETendencyEnable;
   c->SSHTendencyEnable             = P.SSHTendencyEnable;
   c->VelDiffTendencyEnable         = P.VelDiffTendencyEnable;
   c->VelHyperDiffTendencyEnable    = P.VelHyperDiffTendencyEnable;
   c->WindForcingTendencyEnable     = P.WindForcingTendencyEnable;
   c->BottomDragTendencyEnable      = P.BottomDragTendencyEnable;
   c->TracerHorzAdvTendencyEnable   = P.TracerHorzAdvTendencyEnable;
   c->TracerDiffTendencyEnable      = P.TracerDiffTendencyEnable;
   c->TracerHyperDiffTendencyEnable = P.TracerHyperDiffTendencyEnable;
   c->FluxThicknessUpwind           = P.FluxThicknessUpwind;
   c->FluxTracerUpwind              = P.FluxTracerUpwind;
   c->WindInterpIsotropic           = P.WindInterpIsotropic;
   c->ViscDel2 = P.ViscDel2, c->ViscDel4 = P.ViscDel4, c->DivFactor = P.DivFactor;
   c->EddyDiff2 = P.EddyDiff2, c->EddyDiff4 = P.EddyDiff4, c->Density0 = P.Density0;
   c->BottomDragCoeff = P.BottomDragCoeff;
}
static TendParams toParams(const omg_tend_config *c) {
   TendParams P;
   P.ThicknessFluxTendencyEnable   = c->ThicknessFluxTendencyEnable;
   P.PVTendencyEnable              = c->PVTendencyEnable;
   P.KETendencyEnable              = c->KETendencyEnable;
   P.SSHTendencyEnable             = c->SSHTendencyEnable;
   P.VelDiffTendencyEnable         = c->VelDiffTendencyEnable;
   P.VelHyperDiffTendencyEnable    = c->VelHyperDiffTendencyEnable;
   P.WindForcingTendencyEnable     = c->WindForcingTendencyEnable;
   P.BottomDragTendencyEnable      = c->BottomDragTendencyEnable;
   P.TracerHorzAdvTendencyEnable   = c->TracerHorzAdvTendencyEnable;
   P.TracerDiffTendencyEnable      = c->TracerDiffTendencyEnable;
   P.TracerHyperDiffTendencyEnable = c->TracerHyperDiffTendencyEnable;
   P.FluxThicknessUpwind           = c->FluxThicknessUpwind;
   P.FluxTracerUpwind              = c->FluxTracerUpwind;
   P.WindInterpIsotropic           = c->WindInterpIsotropic;
   P.ViscDel2 = c->ViscDel2, P.ViscDel4 = c->ViscDel4, P.DivFactor = c->DivFactor;
   P.EddyDiff2 = c->EddyDiff2, P.EddyDiff4 = c->EddyDiff4, P.Density0 = c->Density0;
   P.BottomDragCoeff = c->BottomDragCoeff;
   return P;
}

// ---------------------------------------------------------------- OceanState
int omg_state_create(const omg_mesh *m, omg_halo *halo, int nvertlayers, int ntimelevels, omg_state **out) {
   OMG_TRY
   OMG_ARG(m && out);
   requireDevice(m->M.get());
   auto *R = new omg_state;
   try {
      R->S.reset(new OceanState("Default", m->M.get(), halo ? halo->H.get() : nullptr, nvertlayers, ntimelevels));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_state_destroy(omg_state *s) {
   delete s;
   return 0;
}
int omg_state_copy_to_device(omg_state *s, int tl, const double *h, const double *u) {
   OMG_TRY
   OMG_ARG(s);
   if (s->S->copyToDevice(h, u, tl) != 0)
      OMEGA_ABORT("OceanState: time level out of range");
   OMG_CATCH
}
int omg_state_copy_to_host(const omg_state *s, int tl, double *h, double *u) {
   OMG_TRY
   OMG_ARG(s);
   if (s->S->copyToHost(h, u, tl) != 0)
      OMEGA_ABORT("OceanState: time level out of range");
   OMG_CATCH
}
int omg_state_device_ptr(const omg_state *s, int tl, int which, double **dev) {
   OMG_TRY
   OMG_ARG(s && dev);
   Array2DReal A;
   const I4 E = which == 0 ? s->S->getLayerThickness(A, tl) : s->S->getNormalVelocity(A, tl);
   if (E != 0)
      OMEGA_ABORT("OceanState: time level out of range");
   *dev = A.Ptr;
   OMG_CATCH
}
int omg_state_exchange_halo(omg_state *s, int tl, void *stream) {
   OMG_TRY
   OMG_ARG(s);
   if (s->S->exchangeHalo(tl, (hipStream_t)stream) != 0)
      OMEGA_ABORT("OceanState::exchangeHalo failed");
   OMG_CATCH
}
int omg_halo_exchange_state(omg_halo *h, omg_state *s, int tl, omg_tracers *t, int tl_tr, void *stream) {
   OMG_TRY
   OMG_ARG(h && s);
   Array2DReal H, U;
   Array3DReal Tr;
   OMEGA_REQUIRE(s->S->getLayerThickness(H, tl) == 0 && s->S->getNormalVelocity(U, tl) == 0,
                 "omg_halo_exchange_state: bad state time level");
   const int NT = t ? t->T->NTracers : 0;
   if (NT > 0)
      OMEGA_REQUIRE(t->T->getAll(Tr, tl_tr) == 0, "omg_halo_exchange_state: bad tracer time level");
   if (h->H->exchangeState(H, U, NT > 0 ? &Tr : nullptr, NT, (hipStream_t)stream) != 0)
      OMEGA_ABORT("Halo::exchangeState failed" + h->H->wireError());
   OMG_CATCH
}
int omg_state_update_time_levels(omg_state *s, void *stream) {
   OMG_TRY
   OMG_ARG(s);
   s->S->updateTimeLevels((hipStream_t)stream);
   OMG_CATCH
}

// ---------------------------------------------------------------- Tracers
int omg_tracers_create(const omg_mesh *m, omg_halo *halo, int nvertlayers, int ntracers, int ntimelevels,
                       omg_tracers **out) {
   OMG_TRY
   OMG_ARG(m && out);
   requireDevice(m->M.get());
   auto *R = new omg_tracers;
   try {
      R->T.reset(new TracerStore(m->M.get(), halo ? halo->H.get() : nullptr, nvertlayers, ntracers, ntimelevels));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_tracers_destroy(omg_tracers *t) {
   delete t;
   return 0;
}
int omg_tracers_copy_to_device(omg_tracers *t, int tl, const double *host) {
   OMG_TRY
   OMG_ARG(t && host);
   if (t->T->copyToDevice(host, tl) != 0)
      OMEGA_ABORT("Tracers: time level out of range");
   OMG_CATCH
}
int omg_tracers_copy_to_host(const omg_tracers *t, int tl, double *host) {
   OMG_TRY
   OMG_ARG(t && host);
   if (t->T->copyToHost(host, tl) != 0)
      OMEGA_ABORT("Tracers: time level out of range");
   OMG_CATCH
}
int omg_tracers_device_ptr(const omg_tracers *t, int tl, double **dev) {
   OMG_TRY
   OMG_ARG(t && dev);
   Array3DReal A;
   if (t->T->getAll(A, tl) != 0)
      OMEGA_ABORT("Tracers: time level out of range");
   *dev = A.Ptr;
   OMG_CATCH
}
int omg_tracers_exchange_halo(omg_tracers *t, int tl, void *stream) {
   OMG_TRY
   OMG_ARG(t);
   if (t->T->exchangeHalo(tl, (hipStream_t)stream) != 0)
      OMEGA_ABORT("Tracers::exchangeHalo failed");
   OMG_CATCH
}
int omg_tracers_update_time_levels(omg_tracers *t, void *stream) {
   OMG_TRY
   OMG_ARG(t);
   t->T->updateTimeLevels((hipStream_t)stream);
   OMG_CATCH
}

// ---------------------------------------------------------------- AuxiliaryState
int omg_aux_create(const omg_mesh *m, omg_halo *halo, int nvertlayers, int ntracers, omg_aux **out) {
   OMG_TRY
   OMG_ARG(m && out);
   requireDevice(m->M.get());
   auto *R = new omg_aux;
   try {
      R->A.reset(new AuxiliaryState("Default", m->M.get(), halo ? halo->H.get() : nullptr, nvertlayers, ntracers));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_aux_destroy(omg_aux *a) {
   delete a;
   return 0;
}
int omg_aux_set_options(omg_aux *a, int ftu, int ftru, int wiso) {
   OMG_TRY
   OMG_ARG(a);
   a->A->LayerThicknessAux.FluxThickEdgeChoice = ftu ? FluxThickEdgeOption::Upwind : FluxThickEdgeOption::Center;
   a->A->TracerAux.TracersOnEdgeChoice         = ftru ? FluxTracerEdgeOption::Upwind : FluxTracerEdgeOption::Center;
   a->A->WindForcingAux.InterpChoice = wiso ? InterpCellToEdgeOption::Isotropic : InterpCellToEdgeOption::Anisotropic;
   OMG_CATCH
}
int omg_aux_compute_mom_aux(omg_aux *a, const omg_state *s, int ttl, int vtl, void *stream) {
   OMG_TRY
   OMG_ARG(a && s);
   a->A->computeMomAux(s->S.get(), ttl, vtl, (hipStream_t)stream);
   OMG_CATCH
}
static Array3DReal tracerArray(const omg_tracers *t, int tl) {
   Array3DReal A;
   OMEGA_REQUIRE(t != nullptr, "tracers handle is NULL");
   if (t->T->getAll(A, tl) != 0)
      OMEGA_ABORT("Tracers: time level out of range");
   return A;
}
int omg_aux_compute_all(omg_aux *a, const omg_state *s, const omg_tracers *t, int trtl, int ttl, int vtl,
                        void *stream) {
   OMG_TRY
   OMG_ARG(a && s);
   a->A->computeAll(s->S.get(), tracerArray(t, trtl), ttl, vtl, (hipStream_t)stream);
   OMG_CATCH
}
static ArrRef auxLookup(const AuxiliaryState &A, const std::string &Name) {
   const std::map<std::string, ArrRef> M{
       {"KineticEnergyCell", arrRef(A.KineticAux.KineticEnergyCell)},
       {"VelocityDivCell", arrRef(A.KineticAux.VelocityDivCell)},
       {"FluxLayerThickEdge", arrRef(A.LayerThicknessAux.FluxLayerThickEdge)},
       {"MeanLayerThickEdge", arrRef(A.LayerThicknessAux.MeanLayerThickEdge)},
       {"SshCell", arrRef(A.LayerThicknessAux.SshCell)},
       {"RelVortVertex", arrRef(A.VorticityAux.RelVortVertex)},
       {"NormRelVortVertex", arrRef(A.VorticityAux.NormRelVortVertex)},
       {"NormPlanetVortVertex", arrRef(A.VorticityAux.NormPlanetVortVertex)},
       {"NormRelVortEdge", arrRef(A.VorticityAux.NormRelVortEdge)},
       {"NormPlanetVortEdge", arrRef(A.VorticityAux.NormPlanetVortEdge)},
       {"Del2Edge", arrRef(A.VelocityDel2Aux.Del2Edge)},
       {"Del2DivCell", arrRef(A.VelocityDel2Aux.Del2DivCell)},
       {"Del2RelVortVertex", arrRef(A.VelocityDel2Aux.Del2RelVortVertex)},
       {"HTracersEdge", arrRef(A.TracerAux.HTracersEdge)},
       {"Del2TracersCell", arrRef(A.TracerAux.Del2TracersCell)},
       {"NormalStressEdge", arrRef(A.WindForcingAux.NormalStressEdge)},
       {"ZonalStressCell", arrRef(A.WindForcingAux.ZonalStressCell)},
       {"MeridStressCell", arrRef(A.WindForcingAux.MeridStressCell)}};
   auto It = M.find(Name);
   if (It == M.end())
      OMEGA_ABORT("AuxiliaryState: no array named " + Name);
   return It->second;
}
int omg_aux_copy_to_host(const omg_aux *a, const char *name, double *host, size_t n) {
   OMG_TRY
   OMG_ARG(a && name && host);
   const ArrRef R = auxLookup(*a->A, name);
   if (n < R.size())
      OMEGA_ABORT(std::string("output buffer too small for ") + name);
   copyRowsToHost(host, R.Ptr, R.Pitch, R.Rows, R.Width);
   OMG_CATCH
}
int omg_aux_copy_to_device(omg_aux *a, const char *name, const double *host, size_t n) {
   OMG_TRY
   OMG_ARG(a && name && host);
   const ArrRef R = auxLookup(*a->A, name);
   if (n != R.size())
      OMEGA_ABORT(std::string("size mismatch for ") + name);
   copyRowsToDevice(R.Ptr, R.Pitch, host, R.Rows, R.Width);
   OMG_CATCH
}
int omg_aux_device_ptr(const omg_aux *a, const char *name, double **dev, size_t *n) {
   OMG_TRY
   OMG_ARG(a && name && dev);
   const ArrRef R = auxLookup(*a->A, name);
   *dev = R.Ptr;
   if (n)
      *n = R.size();
   OMG_CATCH
}

// ---------------------------------------------------------------- Tendencies
static int tendCreate(const omg_mesh *m, int nvertlayers, int ntracers, const omg_tend_config *c, omg_tend **out, bool Allow) {
   OMG_TRY
   OMG_ARG(m && out);
   requireDevice(m->M.get());
   auto *R = new omg_tend;
   try {
      R->T.reset(new Tendencies("Default", m->M.get(), nvertlayers, ntracers, c ? toParams(c) : TendParams(), Allow));
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_tend_create(const omg_mesh *m, int nvertlayers, int ntracers, const omg_tend_config *c, omg_tend **out) {
   return tendCreate(m, nvertlayers, ntracers, c, out, false);
}
int omg_tend_create_reference_structured(const omg_mesh *m, int nvertlayers, int ntracers, const omg_tend_config *c,
                                         omg_tend **out) {
   return tendCreate(m, nvertlayers, ntracers, c, out, true);
}
int omg_tend_fused_limit(int64_t ncells_size, int64_t nedges_size, int64_t nvertices_size, int max_edges, int nvertlayers,
                         int *supported, char *why, size_t why_bytes) {
   OMG_TRY
   OMG_ARG(supported && ncells_size >= 0 && nedges_size >= 0 && nvertices_size >= 0 && nvertlayers > 0);
   const std::string W = Tendencies::fusedLimit((size_t)ncells_size, (size_t)nedges_size, (size_t)nvertices_size, max_edges,
                                                nvertlayers);
   if (why && why_bytes > 0) {
      std::strncpy(why, W.c_str(), why_bytes - 1);
      why[why_bytes - 1] = 0;
   }
   *supported = W.empty() ? 1 : 0;
   OMG_CATCH
}
int omg_tend_destroy(omg_tend *t) {
   delete t;
   return 0;
}
int omg_tend_set_fused(omg_tend *t, int use_fused_rhs) {
   OMG_TRY
   OMG_ARG(t);
   t->T->UseFusedRHS = use_fused_rhs != 0;
   OMG_CATCH
}
int omg_tend_set_graphs(omg_tend *t, int use_graphs) {
   OMG_TRY
   OMG_ARG(t);
   t->T->UseGraphs = use_graphs != 0;
   OMG_CATCH
}
int omg_tend_graph_stats(const omg_tend *t, int64_t *captures, int64_t *replays) {
   OMG_TRY
   OMG_ARG(t);
   if (captures)
      *captures = t->T->Graphs.NCaptures;
   if (replays)
      *replays = t->T->Graphs.NReplays;
   OMG_CATCH
}
int omg_stepper_graph_stats(const omg_stepper *st, int64_t *captures, int64_t *replays) {
   OMG_TRY
   OMG_ARG(st);
   auto *Rk4 = dynamic_cast<RungeKutta4Stepper *>(st->St.get());
   if (captures)
      *captures = Rk4 ? Rk4->Graphs.NCaptures : 0;
   if (replays)
      *replays = Rk4 ? Rk4->Graphs.NReplays : 0;
   OMG_CATCH
}
int omg_tend_compute_all(omg_tend *t, const omg_state *s, omg_aux *a, const omg_tracers *tr, int trtl, int ttl,
                         int vtl, void *stream) {
   OMG_TRY
   OMG_ARG(t && s && a);
   t->T->computeAllTendencies(s->S.get(), a->A.get(), tracerArray(tr, trtl), ttl, vtl, (hipStream_t)stream);
   OMG_CATCH
}
int omg_tend_compute_thickness(omg_tend *t, const omg_state *s, omg_aux *a, int ttl, int vtl, void *stream) {
   OMG_TRY
   OMG_ARG(t && s && a);
   t->T->computeThicknessTendencies(s->S.get(), a->A.get(), ttl, vtl, (hipStream_t)stream);
   OMG_CATCH
}
int omg_tend_compute_velocity(omg_tend *t, const omg_state *s, omg_aux *a, int ttl, int vtl, void *stream) {
   OMG_TRY
   OMG_ARG(t && s && a);
   t->T->computeVelocityTendencies(s->S.get(), a->A.get(), ttl, vtl, (hipStream_t)stream);
   OMG_CATCH
}
int omg_tend_compute_tracer(omg_tend *t, const omg_state *s, omg_aux *a, const omg_tracers *tr, int trtl, int ttl,
                            int vtl, void *stream) {
   OMG_TRY
   OMG_ARG(t && s && a);
   t->T->computeTracerTendencies(s->S.get(), a->A.get(), tracerArray(tr, trtl), ttl, vtl, (hipStream_t)stream);
   OMG_CATCH
}
int omg_tend_compute_thickness_only(omg_tend *t, const omg_state *s, omg_aux *a, int ttl, int vtl, void *stream) {
   OMG_TRY
   OMG_ARG(t && s && a);
   t->T->computeThicknessTendenciesOnly(s->S.get(), a->A.get(), ttl, vtl, (hipStream_t)stream);
   OMG_CATCH
}
int omg_tend_compute_velocity_only(omg_tend *t, const omg_state *s, omg_aux *a, int ttl, int vtl, void *stream) {
   OMG_TRY
   OMG_ARG(t && s && a);
   t->T->computeVelocityTendenciesOnly(s->S.get(), a->A.get(), ttl, vtl, (hipStream_t)stream);
   OMG_CATCH
}
int omg_tend_compute_tracer_only(omg_tend *t, const omg_state *s, omg_aux *a, const omg_tracers *tr, int trtl,
                                 int ttl, int vtl, void *stream) {
   OMG_TRY
   OMG_ARG(t && s && a);
   t->T->computeTracerTendenciesOnly(s->S.get(), a->A.get(), tracerArray(tr, trtl), ttl, vtl, (hipStream_t)stream);
   OMG_CATCH
}
int omg_tend_use_manufactured_solution(omg_tend *t, const omg_mesh *m, double wavelength_x, double wavelength_y,
                                       double amplitude) {
   OMG_TRY
   OMG_ARG(t && m);
   const TendParams &P = t->T->Params;
   auto Ms = std::make_shared<ManufacturedSolution>(m->M.get(), wavelength_x, wavelength_y, amplitude,
                                                    P.VelDiffTendencyEnable != 0, P.VelHyperDiffTendencyEnable != 0,
                                                    P.ViscDel2, P.ViscDel4);
   t->Ms                     = Ms;
   t->T->CustomThicknessTend = [Ms](const Array2DReal &Tend, const OceanState *, const AuxiliaryState *, int, int,
                                    R8 Time, hipStream_t S) { Ms->thicknessTendency(Tend, Time, S); };
   t->T->CustomVelocityTend  = [Ms](const Array2DReal &Tend, const OceanState *, const AuxiliaryState *, int, int,
                                    R8 Time, hipStream_t S) { Ms->velocityTendency(Tend, Time, S); };
   OMG_CATCH
}
int omg_tend_set_custom_tendency(omg_tend *t, int which, omg_custom_tend_fn fn, void *ctx) {
   OMG_TRY
   OMG_ARG(t && (which == 0 || which == 1));
   Tendencies::CustomTendencyType F;
   if (fn) {
      const HorzMesh *M = t->T->Mesh;
      const int NAll = which == 0 ? M->NCellsAll : M->NEdgesAll, NSize = which == 0 ? M->NCellsSize : M->NEdgesSize;
      F = [fn, ctx, NAll, NSize](const Array2DReal &Tend, const OceanState *State, const AuxiliaryState *, int ThickLvl,
                                 int VelLvl, R8 Time, hipStream_t S) {
         Array2DReal H, U;
         OMEGA_REQUIRE(State->getLayerThickness(H, ThickLvl) == 0 && State->getNormalVelocity(U, VelLvl) == 0,
                       "custom tendency: bad time level");
         OMEGA_REQUIRE(fn(ctx, Tend.Ptr, H.Ptr, U.Ptr, NAll, NSize, Tend.Ext[1], Tend.Pitch, Time, (void *)S) == 0,
                       "custom tendency callback reported an error");
      };
   }
   (which == 0 ? t->T->CustomThicknessTend : t->T->CustomVelocityTend) = F;
   OMG_CATCH
}
int omg_tend_clear_custom_tendencies(omg_tend *t) {
   OMG_TRY
   OMG_ARG(t);
   t->T->CustomThicknessTend = nullptr;
   t->T->CustomVelocityTend  = nullptr;
   t->Ms.reset();
   OMG_CATCH
}
int omg_tend_set_time(omg_tend *t, double seconds) {
   OMG_TRY
   OMG_ARG(t);
   t->T->ModelTime = seconds;
   OMG_CATCH
}
int omg_tend_kernel_timing(omg_tend *t, int enable) {
   OMG_TRY
   OMG_ARG(t);
   t->T->enableKernelTiming(enable != 0);
   OMG_CATCH
}
int omg_tend_collect_kernel_times(omg_tend *t, double *ms_sum, int *n_kernels, int *n_samples) {
   OMG_TRY
   OMG_ARG(t && ms_sum && n_kernels && n_samples);
   *n_samples = t->T->collectKernelTimes(ms_sum);
   *n_kernels = FusedNumKernels;
   OMG_CATCH
}
const char *omg_tend_kernel_name(int i) { return (i >= 0 && i < FusedNumKernels) ? FusedKernelNames[i] : ""; }
static ArrRef tendLookup(const Tendencies &T, int Which) {
   switch (Which) {
   case 0:
      return arrRef(T.LayerThicknessTend);
   case 1:
      return arrRef(T.NormalVelocityTend);
   case 2:
      return arrRef(T.TracerTend);
   default:
      OMEGA_ABORT("Tendencies: `which` must be 0, 1 or 2");
   }
}
int omg_tend_copy_to_host(const omg_tend *t, int which, double *host, size_t n) {
   OMG_TRY
   OMG_ARG(t && host);
   const ArrRef R = tendLookup(*t->T, which);
   if (n < R.size())
      OMEGA_ABORT("output buffer too small for tendency array");
   copyRowsToHost(host, R.Ptr, R.Pitch, R.Rows, R.Width);
   OMG_CATCH
}
int omg_tend_device_ptr(const omg_tend *t, int which, double **dev, size_t *n) {
   OMG_TRY
   OMG_ARG(t && dev);
   const ArrRef R = tendLookup(*t->T, which);
   *dev = R.Ptr;
   if (n)
      *n = R.size();
   OMG_CATCH
}
int omg_level_pitch(int nvertlayers) { return levelPitch(nvertlayers); }

// ---------------------------------------------------------------- TimeStepper
int omg_stepper_create(const char *type, double dt, omg_tend *t, omg_aux *a, const omg_mesh *m, omg_halo *halo,
                       omg_tracers *tr, omg_stepper **out) {
   OMG_TRY
   OMG_ARG(type && t && a && m && tr && out);
   const TimeStepperType Ty = TimeStepper::getFromStr(type);
   if (Ty == TimeStepperType::Invalid)
      OMEGA_ABORT(std::string("TimeStepper: unknown type ") + type);
   auto *R = new omg_stepper;
   try {
      R->St.reset(TimeStepper::make("Default", Ty, dt));
      R->St->attachData(t->T.get(), a->A.get(), m->M.get(), halo ? halo->H.get() : nullptr, tr->T.get());
      R->St->finalizeInit();
   } catch (...) {
      delete R;
      throw;
   }
   *out = R;
   OMG_CATCH
}
int omg_stepper_destroy(omg_stepper *st) {
   delete st;
   return 0;
}
int omg_stepper_do_step(omg_stepper *st, omg_state *s, void *stream) {
   OMG_TRY
   OMG_ARG(st && s);
   st->St->doStep(s->S.get(), (hipStream_t)stream);
   OMG_CATCH
}
int omg_stepper_change_time_step(omg_stepper *st, double dt) {
   OMG_TRY
   OMG_ARG(st && dt > 0);
   st->St->changeTimeStep(dt);
   OMG_CATCH
}
int omg_update_by_tend(double *out, const double *in, const double *tend, double coeff, int n_rows, int k, void *stream) {
   OMG_TRY
   OMG_ARG(out && in && tend && n_rows >= 0 && k > 0);
   launchUpdateByTend(n_rows, k, out, in, tend, coeff, (hipStream_t)stream);
   OMG_CATCH
}
int omg_stepper_set_start_time(omg_stepper *st, double seconds) {
   OMG_TRY
   OMG_ARG(st);
   st->St->StartTime = seconds - (double)st->St->NStepsDone * st->St->TimeStepSeconds;
   OMG_CATCH
}
int omg_stepper_get_time(const omg_stepper *st, double *seconds) {
   OMG_TRY
   OMG_ARG(st && seconds);
   *seconds = st->St->simTime();
   OMG_CATCH
}
int omg_stepper_set_option(omg_stepper *st, const char *name, int value) {
   OMG_TRY
   OMG_ARG(st && name);
   auto *Rk4 = dynamic_cast<RungeKutta4Stepper *>(st->St.get());
   const std::string N(name);
   if (Rk4 && N == "FuseStageUpdates")
      Rk4->FuseStageUpdates = value != 0;
   else if (Rk4 && N == "StoreStageTendencies")
      Rk4->StoreStageTendencies = value != 0;
   else if (Rk4 && N == "OverlapHaloExchange")
      Rk4->OverlapHaloExchange = value != 0;
   else if (Rk4 && N == "UseGraphs")
      Rk4->UseGraphs = value != 0;
   else
      OMEGA_ABORT("TimeStepper: no option named " + N + " for this scheme");
   OMG_CATCH
}
int omg_stepper_coeff_seconds(double mult, double dt, double *out) {
   OMG_TRY
   OMG_ARG(out);
   *out = TimeStepper::coeffSeconds(mult, dt);
   OMG_CATCH
}

} // extern "C"
