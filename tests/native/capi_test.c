/* capi_test.c -- a plain C99 consumer of include/omega_amd.h (compiled with gcc, linked with libomega_amd.so):
 * mesh from an MPAS file, Decomp / HorzMesh / OceanState / Tracers / AuxiliaryState / Tendencies / TimeStepper
 * handles, one fused RHS and one RK4 step, compared bit for bit with golden vectors (raw doubles in global order,
 * written by the pytest wrapper from tests/golden/).  usage: capi_test <mesh.nc> <golden_dir> <K> <NT> */
#include "omega_amd.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define OK(call)                                                                                                   \
   do {                                                                                                            \
      if ((call) != 0) {                                                                                           \
         printf("FAIL: %s: %s\n", #call, omg_last_error());                                                        \
         return 1;                                                                                                 \
      }                                                                                                            \
   } while (0)

static double *read_bin(const char *dir, const char *name, size_t n) {
   char path[4096];
   snprintf(path, sizeof path, "%s/%s", dir, name);
   FILE *f = fopen(path, "rb");
   double *v = (double *)malloc(n * sizeof(double));
   if (!f || fread(v, sizeof(double), n, f) != n) {
      printf("FAIL: cannot read %zu doubles from %s\n", n, path);
      exit(1);
   }
   fclose(f);
   return v;
}
/* global [nt][nglobal][k] -> local [nt][rows][k] (last row = zero sentinel) */
static double *to_local(const double *g, const int32_t *id, int rows, int k, int nt, size_t nglobal) {
   double *l = (double *)calloc((size_t)nt * rows * k, sizeof(double));
   for (int t = 0; t < nt; ++t)
      for (int r = 0; r + 1 < rows; ++r)
         memcpy(l + ((size_t)t * rows + r) * k, g + ((size_t)t * nglobal + (id[r] - 1)) * k, k * sizeof(double));
   return l;
}
static int same(const double *l, const double *g, const int32_t *id, int nowned, int rows, int k, int nt, size_t nglobal) {
   for (int t = 0; t < nt; ++t)
      for (int r = 0; r < nowned; ++r)
         if (memcmp(l + ((size_t)t * rows + r) * k, g + ((size_t)t * nglobal + (id[r] - 1)) * k, k * sizeof(double)))
            return 0;
   return 1;
}

int main(int argc, char **argv) {
   if (argc < 5) {
      printf("usage: %s mesh.nc golden_dir K NT\n", argv[0]);
      return 2;
   }
   const char *dir = argv[2];
   const int K = atoi(argv[3]), NT = atoi(argv[4]);
   int fails = 0, ndev = 0;
   OK(omg_device_count(&ndev));
   if (ndev < 1) {
      printf("FAIL: no HIP device (the product has no CPU fallback)\n");
      return 1;
   }
   OK(omg_device_init(0));
   omg_mesh_file *file;
   omg_global_mesh gm;
   OK(omg_mesh_file_open(argv[1], &file));
   OK(omg_mesh_file_global_mesh(file, &gm));
   omg_decomp *decomp;
   omg_mesh *mesh;
   OK(omg_decomp_create(&gm, 1, 0, 3, NULL, &decomp));
   OK(omg_mesh_create(decomp, K, 0, &mesh));
   int32_t ncs, nes, nco, neo;
   OK(omg_mesh_get_int(mesh, "NCellsSize", &ncs));
   OK(omg_mesh_get_int(mesh, "NEdgesSize", &nes));
   OK(omg_mesh_get_int(mesh, "NCellsOwned", &nco));
   OK(omg_mesh_get_int(mesh, "NEdgesOwned", &neo));
   int32_t *cid = (int32_t *)malloc(ncs * sizeof(int32_t)), *eid = (int32_t *)malloc(nes * sizeof(int32_t));
   OK(omg_decomp_get_array(decomp, "CellID", cid, (size_t)ncs));
   OK(omg_decomp_get_array(decomp, "EdgeID", eid, (size_t)nes));
   const size_t ncg = (size_t)gm.nCells, neg = (size_t)gm.nEdges;

   omg_state *state;
   omg_tracers *tracers;
   omg_aux *aux;
   omg_tend *tend;
   omg_tend_config cfg;
   omg_tend_config_default(&cfg);
   OK(omg_state_create(mesh, NULL, K, 2, &state));
   OK(omg_tracers_create(mesh, NULL, K, NT, 2, &tracers));
   OK(omg_aux_create(mesh, NULL, K, NT, &aux));
   OK(omg_tend_create(mesh, K, NT, &cfg, &tend));

   double *hg = read_bin(dir, "h.bin", ncg * K), *ug = read_bin(dir, "u.bin", neg * K);
   double *trg = read_bin(dir, "tr.bin", (size_t)NT * ncg * K);
   double *hl = to_local(hg, cid, ncs, K, 1, ncg), *ul = to_local(ug, eid, nes, K, 1, neg);
   double *tl = to_local(trg, cid, ncs, K, NT, ncg);
   OK(omg_state_copy_to_device(state, 0, hl, ul));
   OK(omg_tracers_copy_to_device(tracers, 0, tl));

   void *stream;
   OK(omg_stream_create(&stream));
   OK(omg_tend_compute_all(tend, state, aux, tracers, 0, 0, 0, stream));
   OK(omg_stream_synchronize(stream));
   double *ht = (double *)malloc((size_t)ncs * K * sizeof(double)), *ut = (double *)malloc((size_t)nes * K * sizeof(double));
   double *tt = (double *)malloc((size_t)NT * ncs * K * sizeof(double));
   OK(omg_tend_copy_to_host(tend, 0, ht, (size_t)ncs * K));
   OK(omg_tend_copy_to_host(tend, 1, ut, (size_t)nes * K));
   OK(omg_tend_copy_to_host(tend, 2, tt, (size_t)NT * ncs * K));
   if (!same(ht, read_bin(dir, "hTend.bin", ncg * K), cid, nco, ncs, K, 1, ncg))
      printf("FAIL: LayerThicknessTend\n"), ++fails;
   if (!same(ut, read_bin(dir, "uTend.bin", neg * K), eid, neo, nes, K, 1, neg))
      printf("FAIL: NormalVelocityTend\n"), ++fails;
   if (!same(tt, read_bin(dir, "trTend.bin", (size_t)NT * ncg * K), cid, nco, ncs, K, NT, ncg))
      printf("FAIL: TracerTend\n"), ++fails;

   omg_stepper *stepper;
   OK(omg_stepper_create("RungeKutta4", 600.0, tend, aux, mesh, NULL, tracers, &stepper));
   OK(omg_stepper_do_step(stepper, state, stream));
   OK(omg_stream_synchronize(stream));
   OK(omg_state_copy_to_host(state, 0, ht, ut));
   OK(omg_tracers_copy_to_host(tracers, 0, tt));
   if (!same(ht, read_bin(dir, "rk4_h.bin", ncg * K), cid, nco, ncs, K, 1, ncg))
      printf("FAIL: RK4 h\n"), ++fails;
   if (!same(ut, read_bin(dir, "rk4_u.bin", neg * K), eid, neo, nes, K, 1, neg))
      printf("FAIL: RK4 u\n"), ++fails;
   if (!same(tt, read_bin(dir, "rk4_tr.bin", (size_t)NT * ncg * K), cid, nco, ncs, K, NT, ncg))
      printf("FAIL: RK4 tracers\n"), ++fails;
   double t = -1;
   OK(omg_stepper_get_time(stepper, &t));
   if (t != 600.0)
      printf("FAIL: stepper time %g\n", t), ++fails;

   /* round 3 entry points, from plain C: options instead of environment variables, the roctx timing level, a halo of
      a one-rank decomposition (nothing travels; every element type must still return 0), a one-rank peer wire
      (mailbox + flags allocated, exported, the rank's own handle block accepted by connect, attached to the halo) */
   {
      int v = -99;
      OK(omg_get_option("MergeL1", &v));
      if (v != 1)
         printf("FAIL: option MergeL1 default %d\n", v), ++fails;
      OK(omg_set_option("MergeL1", 0));
      OK(omg_get_option("MergeL1", &v));
      if (v != 0)
         printf("FAIL: omg_set_option\n"), ++fails;
      OK(omg_set_option("MergeL1", 1));
      if (omg_set_option("NoSuchOption", 1) == 0)
         printf("FAIL: unknown option accepted\n"), ++fails;
      OK(omg_set_timing_level(2));
      omg_halo *halo;
      OK(omg_halo_create(decomp, &halo));
      int nn = -1;
      OK(omg_halo_num_neighbors(halo, &nn));
      if (nn != 0)
         printf("FAIL: one rank has %d neighbours\n", nn), ++fails;
      size_t rows = 99;
      OK(omg_halo_recv_rows(halo, 7, 1, 0, &rows));
      if (rows != 0)
         printf("FAIL: one rank receives %zu rows\n", rows), ++fails;
      void *di4, *dr8;
      OK(omg_device_malloc((size_t)ncs * 3 * sizeof(int32_t), &di4));
      OK(omg_device_malloc((size_t)ncs * sizeof(double), &dr8));
      OK(omg_halo_exchange_i4(halo, (int32_t *)di4, 1, ncs, 3, 0, 0, stream));       /* I4, rank 2 */
      OK(omg_halo_exchange_bytes(halo, dr8, 8, 1, ncs, 1, 1, 0, stream));              /* R8, rank 1 */
      omg_peer *pw;
      char handle[OMG_PEER_HANDLE_BYTES];
      OK(omg_peer_create(1, 0, 4096, &pw));
      OK(omg_peer_local_handle(pw, handle));
      OK(omg_peer_connect(pw, handle));
      OK(omg_peer_set_timeout(pw, 5.0));
      OK(omg_halo_use_peer(halo, pw));
      OK(omg_halo_exchange(halo, (double *)dr8, 1, ncs, 1, 1, 0, stream));
      int64_t nex = -1;
      int st = -1;
      OK(omg_peer_info(pw, &nex, &st));
      if (st != 0)
         printf("FAIL: peer wire status %d\n", st), ++fails;
      OK(omg_stream_synchronize(stream));
      /* round 4 entry points: the wire's verdict after a synchronisation, globalSum over the halo's wire (one task: the
         combination alone), and the resource counter around a step (nothing is created inside doStep) */
      OK(omg_halo_check(halo));
      OK(omg_halo_exchange_state(halo, state, 0, tracers, 0, stream)); /* h, u, tracers: one message per neighbour */
      OK(omg_stream_synchronize(stream));
      OK(omg_halo_check(halo));
      {
         double pairs[4] = {1.0, 1e-17, 3.5, 0.0}, sums[4] = {0, 0, 0, 0};
         OK(omg_halo_global_sum_dd(halo, pairs, 2, sums, stream));
         if (sums[0] != 1.0 || sums[2] != 3.5)
            printf("FAIL: omg_halo_global_sum_dd %g %g\n", sums[0], sums[2]), ++fails;
         int64_t n0 = -1, n1 = -2;
         OK(omg_device_resource_count(&n0));
         OK(omg_stepper_do_step(stepper, state, stream));
         OK(omg_stream_synchronize(stream));
         OK(omg_device_resource_count(&n1));
         if (n0 <= 0 || n0 != n1)
            printf("FAIL: device resources created inside a step: %lld -> %lld\n", (long long)n0, (long long)n1), ++fails;
      }
      OK(omg_halo_destroy(halo));
      OK(omg_peer_destroy(pw));
      OK(omg_device_free(di4));
      OK(omg_device_free(dr8));
   }

   OK(omg_stepper_destroy(stepper));
   OK(omg_tend_destroy(tend));
   OK(omg_aux_destroy(aux));
   OK(omg_tracers_destroy(tracers));
   OK(omg_state_destroy(state));
   OK(omg_stream_destroy(stream));
   OK(omg_mesh_destroy(mesh));
   OK(omg_decomp_destroy(decomp));
   OK(omg_mesh_file_close(file));
   if (!fails)
      printf("capi_test OK\n");
   return fails ? 1 : 0;
}
