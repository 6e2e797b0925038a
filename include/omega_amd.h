/* omega_amd.h -- C ABI of libomega_amd.so, the MI355X-native compute backend for the
 * Omega ocean-dycore hot path (Tendencies + AuxiliaryState + TimeStepper + Halo).
 *
 * Omega has no FFI layer: its "operator API" for this path is a set of C++ classes
 * (O/ = components/omega/ in the reference tree).  Each entry point below names the
 * reference interface it replaces.  The same classes exist in C++ under omega_amd/csrc/
 * (namespace OMEGA, same class and method names) for a source-level drop-in; this C ABI
 * is the boundary a non-C++ host (or a test harness) binds.  See INTEGRATION.md.
 *
 * Conventions
 *  - every function returns 0 on success, non-zero on failure; omg_last_error() returns the
 *    message of the last failure on the calling thread (reference: int return codes of
 *    Halo/OceanState and the ABORT_ERROR macro, O/src/infra/Error.h:207-270);
 *  - arrays are LayoutRight doubles / int32, vertical index innermost, NXxSize = NXxAll+1
 *    rows with a zero sentinel row (O/src/base/DataTypes.h:58-94, O/src/base/Decomp.cpp:1082);
 *  - `stream` arguments are hipStream_t passed as void* (NULL = default stream); every
 *    compute call is asynchronous on that stream;
 *  - no function falls back to the CPU: without a HIP device every device call fails.
 */
#ifndef OMEGA_AMD_H
#define OMEGA_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct omg_decomp omg_decomp;   /* O/src/base/Decomp.h   class Decomp        */
typedef struct omg_halo omg_halo;       /* O/src/base/Halo.h     class Halo          */
typedef struct omg_mesh omg_mesh;       /* O/src/ocn/HorzMesh.h  class HorzMesh      */
typedef struct omg_state omg_state;     /* O/src/ocn/OceanState.h class OceanState   */
typedef struct omg_tracers omg_tracers; /* O/src/ocn/Tracers.h   class Tracers       */
typedef struct omg_aux omg_aux;         /* O/src/ocn/AuxiliaryState.h                */
typedef struct omg_tend omg_tend;       /* O/src/ocn/Tendencies.h                    */
typedef struct omg_stepper omg_stepper; /* O/src/timeStepping/TimeStepper.h          */

enum { OMG_ON_CELL = 0, OMG_ON_EDGE = 1, OMG_ON_VERTEX = 2 }; /* O/src/base/Halo.h:45 MeshElement */

const char *omg_last_error(void);

/* Row pitch of the library's own device arrays whose last index is the vertical level: every array the *_device_ptr
 * entry points return is [rows][omg_level_pitch(K)] doubles of which the first K per row are the levels (columns
 * of at least one 128-byte line are padded to whole lines: K = 60 -> 64; K a multiple of 16 or below 16: pitch = K).
 * Host arrays handed to / returned by the copy entry points and halo messages are always compact [rows][K].
 * Entry points that take RAW device arrays from the caller (omg_halo_exchange, omg_horz_*, omg_update_by_tend,
 * omg_local_weighted_sum_dd) have an explicit row_pitch argument (0 = compact). */
int omg_level_pitch(int nvertlayers);

/* ---- device / stream / event plumbing (Kokkos::initialize, Kokkos::fence, Pacer timers) ---- */
int omg_device_count(int *n);
int omg_device_init(int device_id);
int omg_device_synchronize(void);
int omg_stream_create(void **stream);
int omg_stream_destroy(void *stream);
int omg_stream_synchronize(void *stream);
int omg_event_create(void **event);
int omg_event_destroy(void *event);
int omg_event_record(void *event, void *stream);
int omg_event_elapsed_ms(void *start, void *stop, float *ms); /* synchronises on `stop` */

/* ---- global mesh as read from an MPAS mesh file (O/src/base/Decomp.cpp:108-395 readMesh,
 *      O/src/ocn/HorzMesh.cpp:424-523 read*).  Host pointers; indices 0-based, -1 = missing. ---- */
typedef struct omg_global_mesh {
   int32_t nCells, nEdges, nVertices, maxEdges, vertexDegree;
   const int32_t *cellsOnCell, *edgesOnCell, *verticesOnCell; /* [nCells][maxEdges]      */
   const int32_t *cellsOnEdge, *verticesOnEdge;               /* [nEdges][2]             */
   const int32_t *edgesOnEdge;                                /* [nEdges][2*maxEdges]    */
   const int32_t *cellsOnVertex, *edgesOnVertex;              /* [nVertices][vertexDegree] */
   const double *xCell, *yCell, *zCell, *lonCell, *latCell;
   const double *xEdge, *yEdge, *zEdge, *lonEdge, *latEdge;
   const double *xVertex, *yVertex, *zVertex, *lonVertex, *latVertex;
   const double *areaCell, *areaTriangle, *kiteAreasOnVertex; /* kite: [nVertices][vertexDegree] */
   const double *dcEdge, *dvEdge, *angleEdge, *weightsOnEdge; /* weights: [nEdges][2*maxEdges]   */
   const double *fCell, *fEdge, *fVertex, *bottomDepth;
} omg_global_mesh;

/* ---- raw device buffers for hosts without their own HIP runtime binding (Kokkos::View allocation / deep_copy) ---- */
/* number of device resources (buffers, streams, events) the library has created in this process: everything a time
 * step needs is created by the constructors / omg_stepper_create (O/src/timeStepping/RungeKutta4Stepper.cpp:43-64
 * allocates in finalizeInit), so the number does not move across omg_stepper_do_step calls */
int omg_device_resource_count(int64_t *n);
int omg_device_malloc(size_t bytes, void **ptr);
int omg_device_free(void *ptr);
int omg_copy_to_device(void *dst, const void *src, size_t bytes);
int omg_copy_to_host(void *dst, const void *src, size_t bytes);

/* ---- Reductions (O/src/base/Reductions.h:17-88 and the array forms :150-190, :262-310): sums in double-double.
 *      omg_local_sum_dd: sum a[i] (b == NULL) or sum a[i]*b[i] over n DEVICE values; omg_local_weighted_sum_dd:
 *      sum_r w[r] * sum_k a[r][k](*b[r][k]) over the first nrows rows of [rows][k] device arrays (w per row, e.g.
 *      AreaCell x layer thickness x tracer = tracer content).  hi_lo[0] + hi_lo[1] is the sum.
 *      omg_combine_dd: the reference's MPI_SUMDD operator applied in order to npairs (hi, lo) pairs -- how the
 *      per-rank partial sums are combined after an all-gather (globalSum). ---- */
int omg_local_sum_dd(const double *a, const double *b, size_t n, void *stream, double *hi_lo);
int omg_local_weighted_sum_dd(const double *w, const double *a, const double *b, int nrows, int k, int row_pitch,
                              void *stream, double *hi_lo); /* row_pitch of a and b in values, 0 = k */
int omg_combine_dd(const double *pairs, int npairs, double *hi_lo);

/* ---- MPAS mesh / initial-state file (O/src/base/Decomp.cpp:108-395 readMesh and O/src/ocn/HorzMesh.cpp:424-523
 *      read the same variables through SCORPIO): NetCDF classic CDF-1 / CDF-2 / CDF-5, both name conventions
 *      ("NCells" | "nCells", "CellsOnCell" | "cellsOnCell", ...), indices converted to 0-based / -1.
 *      omg_mesh_file_global_mesh fills `m` with pointers that stay valid until omg_mesh_file_close.
 *      omg_mesh_file_read_f64 reads any variable (e.g. layerThickness, normalVelocity, temperature of an initial
 *      state; record < 0 = all records) converted to double; omg_mesh_file_var_size = its element count
 *      (-1 if absent); omg_mesh_file_dim = a dimension length (-1 if absent). ---- */
typedef struct omg_mesh_file omg_mesh_file;
int omg_mesh_file_open(const char *path, omg_mesh_file **out);
int omg_mesh_file_close(omg_mesh_file *f);
int omg_mesh_file_global_mesh(const omg_mesh_file *f, omg_global_mesh *m);
int omg_mesh_file_dim(const omg_mesh_file *f, const char *name, int64_t *len);
int omg_mesh_file_var_size(const omg_mesh_file *f, const char *name, int64_t record, int64_t *n);
int omg_mesh_file_read_f64(const omg_mesh_file *f, const char *name, int64_t record, double *out, size_t n);

/* ---- Restart file (the RestartWrite / InitialState IOStreams of O/configs/Default.yml:91-127 go through SCORPIO;
 *      here one NetCDF CDF-5 file: layerThickness(nCells,nVertLevels), normalVelocity(nEdges,nVertLevels),
 *      tracers(nTracers,nCells,nVertLevels), simulationTime, stepsDone).  Rank 0 calls omg_restart_create; then
 *      every rank opens the file and writes / reads the rows of its own elements by 1-based global id (CellID /
 *      EdgeID of its Decomp) -- rows are [n][nVertLevels] host doubles; var = "layerThickness" | "normalVelocity" |
 *      "tracers" (plane = tracer index, else 0).  A restart may use a different partition. ---- */
typedef struct omg_restart_file omg_restart_file;
int omg_restart_create(const char *path, int64_t ncells_global, int64_t nedges_global, int nvertlevels, int ntracers,
                       double simulation_time, int64_t steps_done);
int omg_restart_open(const char *path, int write, omg_restart_file **out);
int omg_restart_close(omg_restart_file *f);
int omg_restart_info(const omg_restart_file *f, int64_t *ncells, int64_t *nedges, int *nvertlevels, int *ntracers,
                     double *simulation_time, int64_t *steps_done);
int omg_restart_write_rows(omg_restart_file *f, const char *var, int plane, const int32_t *global_id, int64_t n,
                           const double *rows);
int omg_restart_read_rows(const omg_restart_file *f, const char *var, int plane, const int32_t *global_id, int64_t n,
                          double *rows);

/* ---- History output (the "History" IOStream, O/configs/Default.yml:115-127: `Contents` lists fields and field groups;
 *      field names, long names and units as registered by the registerFields of O/src/ocn/auxiliaryVars/ (one .cpp per group) and
 *      O/src/ocn/OceanState.cpp:190-234).  contents: comma-separated field names ("SshCell", "KineticEnergyCell", ...)
 *      and groups ("State" = LayerThickness + NormalVelocity, "Tracers", "AuxiliaryState" = every auxiliary field).
 *      One CDF-5 file per dump (dimensions NCells / NEdges / NVertices / NVertLayers / NTracers); every rank writes
 *      the rows of its owned elements by global id.  create_file != 0 on exactly one rank (first, then a barrier of
 *      the caller's).  The whole AuxiliaryState is recomputed from the state / tracers at `time_level` before anything
 *      is copied (the fused RHS does not materialise most auxiliary arrays), on `stream`, which is synchronised. ---- */
int omg_history_write(const char *path, const omg_decomp *d, const omg_state *s, const omg_tracers *t, omg_aux *a,
                      const char *contents, double simulation_time, int time_level, int create_file, void *stream,
                      int *n_variables_written);

/* ---- Decomp (O/src/base/Decomp.cpp:444-745 constructor; Decomp.h:189-260 members).
 *      cell_task: optional [nCells] owner task of each cell (e.g. a METIS part file);
 *      NULL = built-in recursive coordinate bisection.  The global mesh arrays must stay
 *      alive until meshes / halos built from the decomp have been created. ---- */
int omg_decomp_create(const omg_global_mesh *mesh, int nparts, int mytask, int halo_width,
                      const int32_t *cell_task, omg_decomp **out);
/* local_order: 0 = the reference's numbering (owned cells in global-id order, every halo layer sorted by global id,
 * O/src/base/Decomp.cpp:1000-1080); 1 = every group ordered along a Morton curve through the cell centres, so that
 * consecutive local elements are spatial neighbours whatever order the mesh file uses; 2 = the same along a Hilbert
 * curve; 3 = k-d order: every group bisected recursively at the median of its widest axis, so that aligned runs of 8 /
 * 16 / 32 local cells -- the kernels' tiles -- are compact patches on the surface the cells live on (spheres: a tile
 * touches 35 distinct cell rows instead of 42 with the 3-D curves).  Edges / vertices follow the cells in every case.
 * Per global id the results of every computation are identical. */
int omg_decomp_create_ordered(const omg_global_mesh *mesh, int nparts, int mytask, int halo_width,
                              const int32_t *cell_task, int local_order, omg_decomp **out);
/* Built-in partitioners of the cell graph (the reference calls METIS_PartGraphKway, O/src/base/Decomp.cpp:868-1000):
 * method "rcb" = recursive coordinate bisection of the cell centres (what omg_decomp_create uses when cell_task is
 * NULL), "graph" = recursive graph bisection with Fiduccia-Mattheyses refinement + greedy k-way refinement on
 * CellsOnCell (no coordinates needed; minimises the edge cut = halo size within a 3 % imbalance tolerance).
 * cell_task_out[nCells] receives the owner task of every cell -- hand it to omg_decomp_create* as cell_task.
 * edge_cut (optional): number of cell-graph edges between different parts. */
int omg_partition_cells(const omg_global_mesh *mesh, int nparts, const char *method, int32_t *cell_task_out,
                        int64_t *edge_cut);
int omg_decomp_destroy(omg_decomp *d);
/* scalar members by reference name: "NCellsOwned", "NCellsAll", "NCellsSize", "NCellsGlobal",
 * "NEdges...", "NVertices...", "MaxEdges", "VertexDegree", "HaloWidth" */
int omg_decomp_get_int(const omg_decomp *d, const char *name, int32_t *out);
/* array members: "CellID" "EdgeID" "VertexID" (1-based global ids, [NXxSize]), "CellLoc"
 * "EdgeLoc" "VertexLoc" ([NXxSize][2]), "NCellsHalo" "NEdgesHalo" "NVerticesHalo"
 * ([HaloWidth]), "CellTask" ([NCellsGlobal]); n = capacity of out in elements */
int omg_decomp_get_array(const omg_decomp *d, const char *name, int32_t *out, size_t n);

/* ---- Halo (O/src/base/Halo.cpp:150-201 constructor, :455-600 exchange lists;
 *      Halo.h:767-915 exchangeFullArrayHalo) ---- */
typedef int (*omg_transport_fn)(void *ctx, int n_neighbors, const int *tasks, void *const *send_ptrs,
                                const size_t *send_bytes, void *const *recv_ptrs, const size_t *recv_bytes,
                                void *stream);
int omg_halo_create(const omg_decomp *d, omg_halo **out);
int omg_halo_destroy(omg_halo *h);
int omg_halo_num_neighbors(const omg_halo *h, int *n);
int omg_halo_neighbor_task(const omg_halo *h, int i, int *task);
int omg_halo_list_size(const omg_halo *h, int i, int elem, int recv, int *n);
int omg_halo_get_list(const omg_halo *h, int i, int elem, int recv, int32_t *out);
int omg_halo_required_bytes(const omg_halo *h, int i, size_t per_cell, size_t per_edge, size_t per_vertex,
                            size_t *bytes);
/* a caller-supplied wire (test rigs: messages staged through the host over gloo / MPI).  The pointers handed to
 * fn are slices of the Halo's own contiguous send / receive buffers in device memory. */
int omg_halo_set_transport(omg_halo *h, omg_transport_fn fn, void *ctx);
/* The production wire: RCCL send / recv over xGMI issued inside the library on the exchange's HIP stream
 * (ncclGroupStart, ncclRecv + ncclSend per neighbour, ncclGroupEnd) -- what the MPI_Irecv / MPI_Isend / MPI_Test
 * sequence of O/src/base/Halo.h:851-907, Halo.cpp:607-757 becomes.  One process per GPU, device selected with
 * omg_device_init first.  Rank 0 calls omg_rccl_get_unique_id and distributes the OMG_RCCL_ID_BYTES bytes by any side
 * channel; omg_rccl_create is collective over all nranks processes (ncclCommInitRank). */
enum { OMG_RCCL_ID_BYTES = 128 };
typedef struct omg_rccl omg_rccl;
int omg_rccl_get_unique_id(char *id /* [OMG_RCCL_ID_BYTES] */);
int omg_rccl_create(const char *id, int nranks, int rank, omg_rccl **out);
int omg_rccl_destroy(omg_rccl *c);
/* ncclCommAbort: give up a communicator (after an error, or when another rank could not create its own); every
 * later exchange on it fails.  An error inside an exchange aborts the communicator by itself. */
int omg_rccl_abort(omg_rccl *c);
/* what RCCL itself reports: communicator size, this rank, library version code, grouped exchanges issued so far */
int omg_rccl_info(const omg_rccl *c, int *nranks, int *rank, int *version, int64_t *exchanges);
/* one grouped exchange on raw device buffers (same argument meaning as omg_transport_fn) */
int omg_rccl_exchange(omg_rccl *c, int n, const int *peers, void *const *send_ptrs, const size_t *send_bytes,
                      void *const *recv_ptrs, const size_t *recv_bytes, void *stream);
/* route this Halo's exchanges through the communicator (which must outlive the Halo's exchanges) */
int omg_halo_use_rccl(omg_halo *h, omg_rccl *c);
/* The second stream-ordered wire: direct peer copies between the GPUs of the node (PeerWire.h).  Every rank owns a
 * mailbox (its receive buffer) and a block of flags in device memory, exported once with hipIpcGetMemHandle; an exchange
 * is pack kernel -> hipMemcpyAsync device-to-device into each neighbour's mailbox -> flag kernels (release store into
 * the peer's flags, bounded acquire wait on the local ones) -> unpack kernel, all on the exchange's stream: what the
 * MPI_Irecv / MPI_Isend / MPI_Test loop of O/src/base/Halo.h:851-907 becomes without any host wait.
 * omg_peer_create after omg_device_init; distribute the OMG_PEER_HANDLE_BYTES of omg_peer_local_handle to all ranks
 * by any side channel, pass all of them in rank order to omg_peer_connect, then omg_halo_use_peer.
 * mailbox_bytes >= omg_halo_recv_rows(...) * k * 8 of the largest exchange. */
enum { OMG_PEER_HANDLE_BYTES = 160 };
typedef struct omg_peer omg_peer;
int omg_peer_create(int nranks, int rank, size_t mailbox_bytes, omg_peer **out);
int omg_peer_destroy(omg_peer *p);
int omg_peer_local_handle(const omg_peer *p, char *out /* [OMG_PEER_HANDLE_BYTES] */);
int omg_peer_connect(omg_peer *p, const char *all_handles /* [nranks][OMG_PEER_HANDLE_BYTES] */);
/* exchanges issued so far; sticky status (0 = fine, else a wait on a peer gave up: 1 = my previous message was never
 * consumed, 2 = a neighbour's message never arrived, 4 = an all-gather of omg_halo_global_sum_dd never got a rank's
 * values; bits may combine); wait bound in seconds.
 * Lifetime: the wire and the Halo it serves (omg_halo_use_peer) may be destroyed in either order -- each releases the
 * other; a Halo whose wire has been destroyed has no wire (its next multi-rank exchange returns an error). */
int omg_peer_info(const omg_peer *p, int64_t *exchanges, int *status);
int omg_peer_set_timeout(omg_peer *p, double seconds);
int omg_halo_use_peer(omg_halo *h, omg_peer *p);
/* rows of k values this rank receives in one exchange of per_cell / per_edge / per_vertex arrays per element */
int omg_halo_recv_rows(const omg_halo *h, size_t per_cell, size_t per_edge, size_t per_vertex, size_t *rows);
/* Halo::exchangeFullArrayHalo on a raw device array [nt][rows_size][row_pitch] (nt = 1 for 2-D) of which the first
 * k values of every row are exchanged (row_pitch 0 = compact rows of k) */
int omg_halo_exchange(omg_halo *h, double *dev_array, int nt, int rows_size, int k, int row_pitch, int elem,
                      void *stream);
/* the same for the reference's other array types (O/src/base/Halo.h:304-351 rank 1, :324-760 I4 / I8 / R4 / R8, ranks
 * 1-5): values of elem_bytes bytes (4 or 8).  Rank 1: nt = 1, k = 1, row_pitch = 1; ranks 4 and 5: nt = the product of
 * the leading extents (the element index is the second-to-last, Halo.h:418-470).  omg_halo_exchange_i4 = elem_bytes 4. */
int omg_halo_exchange_bytes(omg_halo *h, void *dev_array, int elem_bytes, int nt, int rows_size, int k, int row_pitch,
                            int elem, void *stream);
int omg_halo_exchange_i4(omg_halo *h, int32_t *dev_array, int nt, int rows_size, int k, int row_pitch, int elem,
                         void *stream);
/* globalSum (O/src/base/Reductions.h:71-88 scalar, :150-190 array forms: MPI_Allreduce with the double-double operator
 * MPI_SUMDD) for npairs (<= 64) local partial sums at once: local_hi_lo[npairs][2] (from omg_local_sum_dd /
 * omg_local_weighted_sum_dd) is all-gathered over the Halo's wire -- ncclAllGather on the RCCL communicator, or the
 * peer wire's gather slots -- and combined in rank order with omg_combine_dd: hi_lo[npairs][2], the same bits on every
 * rank and for every partition.  Collective over all tasks of the decomposition; synchronises `stream`.  One task: the
 * combination alone.  Fails on a Halo whose wire is a caller-supplied transport. */
int omg_halo_global_sum_dd(omg_halo *h, const double *local_hi_lo, int npairs, double *hi_lo, void *stream);
/* h, u and the tracers of one exchange point as ONE message per neighbour: what the time steppers do after a stage
 * (RungeKutta4Stepper.cpp:95-101 calls OceanState::exchangeHalo and Tracers::exchangeHalo -- three rounds of messages;
 * here [h on cells][u on edges][tracers on cells] travel together).  `tracers` may be NULL.  Queued on `stream`. */
int omg_halo_exchange_state(omg_halo *h, omg_state *state, int state_time_level, omg_tracers *tracers,
                            int tracers_time_level, void *stream);
/* The wire's verdict, to be asked after the host has synchronised with an exchange's stream (the exchange calls return
 * when the work is queued): 0 = fine; fails (message: omg_last_error) when a peer-wire wait gave up -- the unpack
 * kernel of that exchange then copied nothing, the halo is stale.  The time steppers ask at the start of every step. */
int omg_halo_check(const omg_halo *h);

/* ---- Test options (omega_amd/csrc/Tuning.h).  The library never reads the environment.  Defaults are what production
 *      runs; every option forces a kernel / table structure that some mesh class reaches by itself (named in Tuning.h), so
 *      that tests can drive generated meshes through it -- no option can make a result wrong, and there is no measurement
 *      probe in the library.  Names: MergeL1 Pair TracerPatch (structure of the fused RHS; read at every launch); SendBand
 *      BandOnComm ShrinkSweeps (what a rank leaves out inside an RK4 step; read at every stage); ForceGeneric KeepMaxEdges
 *      NarrowTables (mesh tables; read when a HorzMesh is created); Graphs (-1 per object, 0 never, 1 default on).
 *      Unknown names fail.
 *      omg_set_timing_level: roctx ranges named after the reference's Pacer timers ("Tend:...", "AuxState:...",
 *      "RK4:haloExch"; share/pacer/Pacer.cpp:138-200) are emitted for timers up to this level (default 3 = all). ---- */
int omg_set_option(const char *name, int value);
int omg_get_option(const char *name, int *value);
int omg_set_timing_level(int level);

/* ---- HorzMesh (O/src/ocn/HorzMesh.cpp:44-140 constructor; HorzMesh.h:100-265 members).
 *      host_only != 0 builds the host arrays only (no device mirrors; compute calls fail). ---- */
int omg_mesh_create(const omg_decomp *d, int nvertlayers, int host_only, omg_mesh **out);
int omg_mesh_destroy(omg_mesh *m);
int omg_mesh_get_int(const omg_mesh *m, const char *name, int32_t *out);
int omg_mesh_get_array_i4(const omg_mesh *m, const char *name, int32_t *out, size_t n);
int omg_mesh_get_array_r8(const omg_mesh *m, const char *name, double *out, size_t n);
int omg_mesh_set_fvertex(omg_mesh *m, const double *host_values /* [NVerticesSize] */);

/* ---- HorzOperators (O/src/ocn/HorzOperators.h:9-187; constructors HorzOperators.cpp:7-28): the reference's
 *      reusable operators as sweeps over elements [0, n) (n < 0: all local elements) x nvertlayers levels of raw
 *      device arrays [NXxSize][row_pitch] (row_pitch 0 = compact rows of nvertlayers; input and output share it):
 *        divergence        DivCell(i,k)   = -sum_j DvEdge*EdgeSignOnCell(i,j)*VecEdge(e_j,k)/AreaCell(i)        :13-33
 *        gradient          GradEdge(e,k)  = (Scalar(c1,k) - Scalar(c0,k))/DcEdge(e)                             :47-60
 *        curl              CurlVertex(v,k)= sum_j DcEdge*EdgeSignOnVertex(v,j)*VecEdge(e_j,k)/AreaTriangle(v)   :71-93
 *        tangential_recon  ReconEdge(e,k) = sum_j WeightsOnEdge(e,j)*VecEdge(EdgesOnEdge(e,j),k)                :107-126
 *        interp_cell_to_edge (1-D arrays) isotropic != 0: kite-area weighted over the cells of the edge's two
 *                          vertices (:161-180), else the mean of its two cells (:153-159) ---- */
int omg_horz_divergence(const omg_mesh *m, const double *vec_edge_dev, double *div_cell_dev, int nvertlayers,
                        int row_pitch, int n, void *stream);
int omg_horz_gradient(const omg_mesh *m, const double *scalar_cell_dev, double *grad_edge_dev, int nvertlayers,
                      int row_pitch, int n, void *stream);
int omg_horz_curl(const omg_mesh *m, const double *vec_edge_dev, double *curl_vertex_dev, int nvertlayers,
                  int row_pitch, int n, void *stream);
int omg_horz_tangential_recon(const omg_mesh *m, const double *vec_edge_dev, double *recon_edge_dev, int nvertlayers,
                              int row_pitch, int n, void *stream);
int omg_horz_interp_cell_to_edge(const omg_mesh *m, const double *array_cell_dev, double *array_edge_dev,
                                 int isotropic, int n, void *stream);

/* ---- options (O/configs/Default.yml:25-52; Tendencies::readTendConfig O/src/ocn/Tendencies.cpp:123-213;
 *      AuxiliaryState::readConfigOptions O/src/ocn/AuxiliaryState.cpp:259-308) ---- */
typedef struct omg_tend_config {
   int32_t ThicknessFluxTendencyEnable, PVTendencyEnable, KETendencyEnable, SSHTendencyEnable,
       VelDiffTendencyEnable, VelHyperDiffTendencyEnable, WindForcingTendencyEnable, BottomDragTendencyEnable,
       TracerHorzAdvTendencyEnable, TracerDiffTendencyEnable, TracerHyperDiffTendencyEnable;
   int32_t FluxThicknessUpwind, FluxTracerUpwind, WindInterpIsotropic;
   double ViscDel2, ViscDel4, DivFactor, EddyDiff2, EddyDiff4, Density0, BottomDragCoeff;
} omg_tend_config;
void omg_tend_config_default(omg_tend_config *c);

/* ---- OceanState (O/src/ocn/OceanState.h:100-149; OceanState.cpp:247-407).  time_level:
 *      1 = new, 0 = current, -1 = previous.  halo may be NULL on a single rank. ---- */
int omg_state_create(const omg_mesh *m, omg_halo *halo, int nvertlayers, int ntimelevels, omg_state **out);
int omg_state_destroy(omg_state *s);
int omg_state_copy_to_device(omg_state *s, int time_level, const double *h_host, const double *u_host);
int omg_state_copy_to_host(const omg_state *s, int time_level, double *h_host, double *u_host);
int omg_state_device_ptr(const omg_state *s, int time_level, int which /*0 h, 1 u*/, double **dev);
int omg_state_exchange_halo(omg_state *s, int time_level, void *stream);
int omg_state_update_time_levels(omg_state *s, void *stream);

/* ---- Tracers (O/src/ocn/Tracers.cpp:269 getAll, :457-496 exchangeHalo/updateTimeLevels) ---- */
int omg_tracers_create(const omg_mesh *m, omg_halo *halo, int nvertlayers, int ntracers, int ntimelevels,
                       omg_tracers **out);
int omg_tracers_destroy(omg_tracers *t);
int omg_tracers_copy_to_device(omg_tracers *t, int time_level, const double *host);
int omg_tracers_copy_to_host(const omg_tracers *t, int time_level, double *host);
int omg_tracers_device_ptr(const omg_tracers *t, int time_level, double **dev);
int omg_tracers_exchange_halo(omg_tracers *t, int time_level, void *stream);
int omg_tracers_update_time_levels(omg_tracers *t, void *stream);

/* ---- AuxiliaryState (O/src/ocn/AuxiliaryState.h:49-82; AuxiliaryState.cpp:60-191) ---- */
int omg_aux_create(const omg_mesh *m, omg_halo *halo, int nvertlayers, int ntracers, omg_aux **out);
int omg_aux_destroy(omg_aux *a);
int omg_aux_set_options(omg_aux *a, int flux_thickness_upwind, int flux_tracer_upwind, int wind_interp_isotropic);
int omg_aux_compute_mom_aux(omg_aux *a, const omg_state *s, int thick_time_level, int vel_time_level, void *stream);
int omg_aux_compute_all(omg_aux *a, const omg_state *s, const omg_tracers *t, int tracer_time_level,
                        int thick_time_level, int vel_time_level, void *stream);
/* array by reference member name ("KineticEnergyCell", "VelocityDivCell", "FluxLayerThickEdge",
 * "MeanLayerThickEdge", "SshCell", "RelVortVertex", "NormRelVortVertex", "NormPlanetVortVertex",
 * "NormRelVortEdge", "NormPlanetVortEdge", "Del2Edge", "Del2DivCell", "Del2RelVortVertex",
 * "HTracersEdge", "Del2TracersCell", "NormalStressEdge", "ZonalStressCell", "MeridStressCell") */
int omg_aux_copy_to_host(const omg_aux *a, const char *name, double *host, size_t n);
int omg_aux_copy_to_device(omg_aux *a, const char *name, const double *host, size_t n);
int omg_aux_device_ptr(const omg_aux *a, const char *name, double **dev, size_t *n);

/* ---- Tendencies (O/src/ocn/Tendencies.h:73-102; Tendencies.cpp:257-600) ---- */
/* omg_tend_create FAILS (message: omg_last_error, naming the limit) for a mesh outside the fused RHS: an array plane of
 * 4 GiB or more (32-bit buffer offsets: ~ 2.2 M cells x 80 levels per rank -- several ranks may share a GPU) or MaxEdges
 * outside 5..8.  omg_tend_create_reference_structured accepts such a mesh: computeAllTendencies then takes the
 * reference-structured 23-launch path (~ 5 x the time) -- the caller's decision, not a surprise.
 * omg_tend_fused_limit asks beforehand, on sizes alone (rows incl. the sentinel; no mesh, no device): *supported = 1, or 0
 * with the reason in `why`. */
int omg_tend_create(const omg_mesh *m, int nvertlayers, int ntracers, const omg_tend_config *c, omg_tend **out);
int omg_tend_create_reference_structured(const omg_mesh *m, int nvertlayers, int ntracers, const omg_tend_config *c,
                                         omg_tend **out);
int omg_tend_fused_limit(int64_t ncells_size, int64_t nedges_size, int64_t nvertices_size, int max_edges, int nvertlayers,
                         int *supported, char *why, size_t why_bytes);
int omg_tend_destroy(omg_tend *t);
int omg_tend_set_fused(omg_tend *t, int use_fused_rhs);
/* HIP-graph replay of launch-bound sequences (default off: measured without gain on MI355X; OMEGA_GRAPHS=1 or these
 * switches turn it on): the fused RHS (omg_tend_compute_all) and, on one rank, the
 * stage-fused RK4 step (omg_stepper_do_step; option "UseGraphs") are captured the second time they are called with
 * the same arrays on the same NON-default stream and replayed afterwards with one host call.  Kernel timing and
 * custom tendencies switch it off.  *_graph_stats: graphs captured / replays so far. */
int omg_tend_set_graphs(omg_tend *t, int use_graphs);
int omg_tend_graph_stats(const omg_tend *t, int64_t *captures, int64_t *replays);
int omg_stepper_graph_stats(const omg_stepper *st, int64_t *captures, int64_t *replays);
int omg_tend_compute_all(omg_tend *t, const omg_state *s, omg_aux *a, const omg_tracers *tr, int tracer_time_level,
                         int thick_time_level, int vel_time_level, void *stream);
int omg_tend_compute_thickness(omg_tend *t, const omg_state *s, omg_aux *a, int thick_time_level,
                               int vel_time_level, void *stream);
int omg_tend_compute_velocity(omg_tend *t, const omg_state *s, omg_aux *a, int thick_time_level, int vel_time_level,
                              void *stream);
int omg_tend_compute_tracer(omg_tend *t, const omg_state *s, omg_aux *a, const omg_tracers *tr,
                            int tracer_time_level, int thick_time_level, int vel_time_level, void *stream);
int omg_tend_compute_thickness_only(omg_tend *t, const omg_state *s, omg_aux *a, int thick_time_level,
                                    int vel_time_level, void *stream);
int omg_tend_compute_velocity_only(omg_tend *t, const omg_state *s, omg_aux *a, int thick_time_level,
                                   int vel_time_level, void *stream);
int omg_tend_compute_tracer_only(omg_tend *t, const omg_state *s, omg_aux *a, const omg_tracers *tr,
                                 int tracer_time_level, int thick_time_level, int vel_time_level, void *stream);
/* per-kernel timing of the fused RHS with HIP events on the launch stream (Pacer "Tend:*" timers,
 * O/src/ocn/Tendencies.cpp:280-481).  names[i] / ms_sum[i] for i < *n_kernels (<= 8). */
/* Custom tendencies (O/src/ocn/Tendencies.h:51-53; Tendencies.cpp:41-64 UseCustomTendency +
 * ManufacturedSolutionTendency): attach the manufactured-solution source terms of
 * O/src/ocn/CustomTendencyTerms.cpp:18-208 (wavelengths / amplitude = the ManufacturedSolution config group,
 * Default.yml:143-146; H0 = BottomDepth of the first cell; del2 / del4 switches and viscosities from this
 * Tendencies' options).  omg_tend_set_time = the TimeInstant argument of the compute* calls, in seconds
 * since the reference time (the time steppers set it per stage). */
int omg_tend_use_manufactured_solution(omg_tend *t, const omg_mesh *m, double wavelength_x, double wavelength_y,
                                       double amplitude);
/* Custom tendencies as a caller-supplied function (Tendencies::CustomTendencyType, O/src/ocn/Tendencies.h:51-53:
 * std::function<void(Array2DReal Tend, const OceanState*, const AuxiliaryState*, int ThickTimeLevel,
 * int VelTimeLevel, TimeInstant)>, called at the end of the thickness / velocity group, Tendencies.cpp:288-291,
 * 416-419).  The callback ADDS its term to tend_dev ([n_rows_size][row_pitch] device doubles, the first
 * nvertlayers values of rows [0, n_rows_all) are to be written) on `stream`; it is handed the state the tendencies
 * are evaluated on as raw device arrays of the same row pitch (layer thickness [NCellsSize][row_pitch] at the
 * thickness time level, normal velocity [NEdgesSize][row_pitch] at the velocity time level) and the model time.  which: 0 thickness, 1 velocity.  fn == NULL clears
 * that hook.  With a custom hook set the Runge-Kutta stage updates run as separate kernels. */
typedef int (*omg_custom_tend_fn)(void *ctx, double *tend_dev, const double *layer_thickness_dev,
                                  const double *normal_velocity_dev, int n_rows_all, int n_rows_size, int nvertlayers,
                                  int row_pitch, double time_seconds, void *stream);
int omg_tend_set_custom_tendency(omg_tend *t, int which, omg_custom_tend_fn fn, void *ctx);
int omg_tend_clear_custom_tendencies(omg_tend *t);
int omg_tend_set_time(omg_tend *t, double seconds);
int omg_tend_kernel_timing(omg_tend *t, int enable);
int omg_tend_collect_kernel_times(omg_tend *t, double *ms_sum, int *n_kernels, int *n_samples);
const char *omg_tend_kernel_name(int i);
/* which: 0 LayerThicknessTend [NCellsSize][K], 1 NormalVelocityTend [NEdgesSize][K],
 *        2 TracerTend [NT][NCellsSize][K] */
int omg_tend_copy_to_host(const omg_tend *t, int which, double *host, size_t n);
int omg_tend_device_ptr(const omg_tend *t, int which, double **dev, size_t *n);

/* ---- TimeStepper (O/src/timeStepping/TimeStepper.h:57-139 create/doStep;
 *      RungeKutta4Stepper.cpp:68-137, RungeKutta2Stepper.cpp:27-73, ForwardBackwardStepper.cpp:27-82).
 *      type: "Forward-Backward" | "RungeKutta4" | "RungeKutta2" (TimeStepper.h:64-75) ---- */
int omg_stepper_create(const char *type, double time_step_seconds, omg_tend *t, omg_aux *a, const omg_mesh *m,
                       omg_halo *halo, omg_tracers *tr, omg_stepper **out);
int omg_stepper_destroy(omg_stepper *st);
int omg_stepper_do_step(omg_stepper *st, omg_state *s, void *stream);
/* model time in seconds since the reference time (the reference's SimTime / StepClock); doStep hands the
 * stage times to the custom tendencies */
int omg_stepper_set_start_time(omg_stepper *st, double seconds);
int omg_stepper_get_time(const omg_stepper *st, double *seconds);
/* RungeKutta4 only: "FuseStageUpdates" (default 1: the stage updates of TimeStepper.cpp:378-524 run in the
 * epilogue of the RHS kernels, same arithmetic) and "StoreStageTendencies" (default 0: with fused stages the
 * Tendencies arrays are not written), "OverlapHaloExchange" (default 1: with fused stages and neighbours, each
 * exchange starts when the band of cells whose values travel is final and runs on a communication stream
 * while the stage's interior cells are computed), "UseGraphs" (default 0, see omg_tend_set_graphs).  0 / 1. */
int omg_stepper_set_option(omg_stepper *st, const char *name, int value);
/* TimeStepper::changeTimeStep (O/src/timeStepping/TimeStepper.h:141-143) */
int omg_stepper_change_time_step(omg_stepper *st, double time_step_seconds);
/* the update kernel of TimeStepper::updateThicknessByTend / updateVelocityByTend (O/src/timeStepping/
 * TimeStepper.cpp:378-441) on raw device arrays: out[r][k] = in[r][k] + coeff * tend[r][k] for r < n_rows
 * (out may alias in) */
int omg_update_by_tend(double *out_dev, const double *in_dev, const double *tend_dev, double coeff, int n_rows,
                       int nvertlayers, void *stream);
int omg_stepper_coeff_seconds(double mult, double time_step_seconds, double *out);

#ifdef __cplusplus
}
#endif
#endif
