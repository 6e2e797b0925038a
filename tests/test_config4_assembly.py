"""BASELINE configs[4] -- the oRRS18to6-sized mesh (planar 1924 x 1924 = 3 701 776 cells), 80 levels, 37 tracers, 8 GPUs --
assembled at its stated size: the 8-way graph partition, every rank's Decomp and Halo on the host (no device), and ONE real
rank of it on the GPU (halo included, 37 tracers, ~ 60 GB of arrays) with a wire that delivers the halo's initial-state rows.

The reference reads and partitions the mesh in a distributed way and initialises the state per task
(components/omega/src/base/Decomp.cpp:108-395 readMesh, :868-1000 partition + scatter; src/ocn/OceanState.cpp:65-117);
here every rank holds the global connectivity (5.5 GB of host memory at this size) and builds only its own rows of the
state (meshgen.synthetic_state_rows)."""
import numpy as np
import pytest

import omega_amd as oa
from omega_amd.meshgen import planar_hex, synthetic_state_rows

NX, K, NT, NPARTS, HALO = 1924, 80, 37, 8, 4


@pytest.fixture(scope="module")
def big():
    g = planar_hex(NX, NX, 6.0e3)
    gm = oa.GlobalMesh(g)
    cell_task, cut = oa.partition_cells(gm, NPARTS, "graph")
    return g, gm, cell_task, cut


def test_eight_way_partition_decomp_and_halo_of_the_full_mesh_on_the_host(big):
    g, gm, cell_task, cut = big
    sizes = np.bincount(cell_task, minlength=NPARTS)
    assert sizes.min() > 0 and sizes.max() <= 1.03 * g["nCells"] / NPARTS, sizes   # the partitioner's 3 % tolerance
    assert cut < 4 * 8 * NX                                                          # a few straight cuts' worth of edges
    own_sum = np.zeros(3, dtype=object)
    for r in range(NPARTS):
        d = oa.Decomp(gm, NPARTS, r, HALO, cell_task=cell_task, local_order="kd")
        for i, (arr, n) in enumerate((("CellID", "NCellsOwned"), ("EdgeID", "NEdgesOwned"), ("VertexID", "NVerticesOwned"))):
            own_sum[i] += int(d.get_array(arr)[: d.get_int(n)].astype(np.int64).sum())
        h = oa.Halo(d)
        nbrs = h.neighbors
        assert 1 <= len(nbrs) <= 32, nbrs                                            # PeerWire::MaxPeers
        rows = h.recv_rows(1 + NT, 1, 0)                                             # h + 37 tracers on cells, u on edges
        mailbox = rows * K * 8
        assert 0 < mailbox < 1 << 30, mailbox                                        # ~ 280 MB per rank: one hipMalloc, one IPC handle
        # 32-bit job table of the pack / unpack kernels (Halo.cpp: plane = tracer * RowsSize + row) and the fused RHS's
        # 32-bit byte offsets inside one array plane (FusedKernelsImpl.h: BufOOB)
        assert NT * (d.get_int("NCellsAll") + 1) < 1 << 31
        assert (d.get_int("NEdgesAll") + 1) * K * 8 < 0xffffff00
        assert d.get_int("NCellsAll") < 1.05 * d.get_int("NCellsOwned")             # HaloWidth 4: ~ 2.3 % more cells
    n = [g["nCells"], g["nEdges"], g["nVertices"]]
    assert [int(x) for x in own_sum] == [m * (m + 1) // 2 for m in n]              # every element owned exactly once


@pytest.mark.gpu
def test_one_real_rank_of_the_eight_on_the_gpu(big):
    g, gm, cell_task, _ = big
    oa.device_init(0)
    rank = 3
    d = oa.Decomp(gm, NPARTS, rank, HALO, cell_task=cell_task, local_order="kd")
    mesh = oa.HorzMesh(d, K)
    halo = oa.Halo(d)
    for f in ("CellL1OK", "CellPVOK", "CellPVFinalOK", "Del2RingOK", "Del2VertOK"):
        assert mesh.get_int(f) == 1
    cells0 = d.get_array("CellID")[: mesh.NCellsAll] - 1
    edges0 = d.get_array("EdgeID")[: mesh.NEdgesAll] - 1
    kp = oa.level_pitch(K)
    hh, uu, _ = synthetic_state_rows(g, K, 0, cells0, edges0, tracers=[])
    h = np.zeros((mesh.NCellsSize, K)); h[: mesh.NCellsAll] = hh
    u = np.zeros((mesh.NEdgesSize, K)); u[: mesh.NEdgesAll] = uu
    state = oa.OceanState(mesh, halo, K, 2)
    tracers = oa.Tracers(mesh, halo, K, NT, 2)
    aux = oa.AuxiliaryState(mesh, halo, K, NT)
    tend = oa.Tendencies(mesh, K, NT, oa.default_config())
    state.copy_to_device(h, u, 0)

    # the rank's 37 tracers synthesised ONCE on the host (11 GB, in the device's row pitch; bench.py, which has no reason to
    # upload them six times, builds and uploads one tracer at a time)
    tr_host = np.zeros((NT, mesh.NCellsSize, kp))
    for l in range(NT):
        tr_host[l, : mesh.NCellsAll, :K] = synthetic_state_rows(g, K, NT, cells0, edges0[:0], tracers=[l])[2][0]

    def upload(scale=1.0):
        for l in range(NT):
            oa.copy_to_device(tracers.device_ptr(0) + 8 * l * mesh.NCellsSize * kp, tr_host[l] if scale == 1.0 else tr_host[l] * scale)
    upload()
    nc, ne = mesh.NCellsOwned, mesh.NEdgesOwned
    tend.compute_all_tendencies(state, aux, tracers)
    oa.device_synchronize()
    hT, uT = tend.get(0)[:nc].copy(), tend.get(1)[:ne].copy()
    trT = tend.get(2)[[0, 17, 36], :nc].copy()
    assert np.isfinite(hT).all() and np.isfinite(uT).all() and np.isfinite(trT).all() and np.abs(trT).max() > 0
    # ---- the oracle at this rank's size (r6).  Tracers do not couple in the RHS (TendencyTerms.h:343-492: every tracer
    # term reads h, u and ITS tracer), so the oracle runs the rank's local mesh with h, u and tracers {0, 17, 36} as an
    # NT = 3 problem (~ one headline-size oracle evaluation): hTend, uTend and those three planes of the NT = 37 GPU
    # evaluation must be bit-identical on owned elements.
    from oracle import oracle as O
    SUB = [0, 17, 36]
    omesh = O.Mesh(mesh.local_arrays(), K)
    orc = O.Oracle(omesh, len(SUB), O.default_config())
    tr3 = np.ascontiguousarray(tr_host[SUB][:, :, :K])
    ohT, ouT, otrT = orc.compute_all_tendencies(h, u, tr3)
    assert np.array_equal(hT, ohT[:nc]), "configs[4] rank: LayerThicknessTend differs from the oracle"
    assert np.array_equal(uT, ouT[:ne]), "configs[4] rank: NormalVelocityTend differs from the oracle"
    assert np.array_equal(trT, otrT[:, :nc]), "configs[4] rank: TracerTend planes 0 / 17 / 36 differ from the oracle"
    del ohT, ouT, otrT
    # determinism, and the reference-structured launch sequence gives the same bits on owned elements
    tend.compute_all_tendencies(state, aux, tracers)
    oa.device_synchronize()
    assert np.array_equal(tend.get(0)[:nc], hT) and np.array_equal(tend.get(1)[:ne], uT)
    tend.set_fused(False)
    tend.compute_all_tendencies(state, aux, tracers)
    oa.device_synchronize()
    assert np.array_equal(tend.get(0)[:nc], hT) and np.array_equal(tend.get(1)[:ne], uT)
    assert np.array_equal(tend.get(2)[[0, 17, 36], :nc], trT)
    tend.set_fused(True)
    # the tracer tendency is linear in the tracer: a power of two scales it exactly
    upload(4.0)
    tend.compute_all_tendencies(state, aux, tracers)
    oa.device_synchronize()
    assert np.array_equal(tend.get(2)[[0, 17, 36], :nc], 4.0 * trT)
    upload()

    # ---- two RK4 steps of the rank, with a wire that can be checked -------------------------------------------------
    # The other seven ranks do not exist here, but every row of the initial state is a function of its global id: the
    # wire below delivers, at EVERY exchange, the initial-state rows of this rank's halo elements (a halo frozen at the
    # initial state: a Dirichlet rim around the part) in the message layout of the real exchange -- per neighbour
    # [h on its cell list][u on its edge list][37 tracers on its cell list], K values per row (Halo.cpp: planFor; the
    # reference's layout, Halo.h:344-351).  The receive buffer is written once per buffer address (synchronous copy,
    # complete before the unpack kernel is queued); every later exchange finds the same bytes.  With that, the band /
    # interior split, the pack and unpack of a 280 MB exchange, 32-bit plane offsets at 37 tracers and the job table at
    # this size are all on the path of a result that can be compared: every owned value must stay finite, overlapped
    # exchanges must give the bits of sequential ones, and stage-fused kernels the bits of the reference-structured
    # launch sequence.
    nb = halo.neighbors
    msgs = []
    for i in range(len(nb)):
        lc, le = halo.get_list(i, 0, True), halo.get_list(i, 1, True)
        assert lc.min() >= nc and le.min() >= ne                    # halo elements only
        hh_i, uu_i, _ = synthetic_state_rows(g, K, 0, cells0[lc], edges0[le], tracers=[])
        parts = [hh_i, uu_i]
        for l in range(NT):
            parts.append(synthetic_state_rows(g, K, NT, cells0[lc], edges0[:0], tracers=[l])[2][0])
        msgs.append(np.ascontiguousarray(np.concatenate(parts, axis=0)))
    filled, calls = set(), []

    def frozen_halo_wire(tasks, send_ptrs, send_bytes, recv_ptrs, recv_bytes, stream_handle):
        assert list(tasks) == nb
        for i, m in enumerate(msgs):
            if recv_bytes[i] != m.nbytes:
                raise RuntimeError(f"neighbour {tasks[i]}: the exchange expects {recv_bytes[i]} bytes, the state message has {m.nbytes}")
            if recv_ptrs[i] not in filled:
                oa.copy_to_device(recv_ptrs[i], m)
                filled.add(recv_ptrs[i])
        calls.append(sum(send_bytes))
        return 0
    halo.set_transport(frozen_halo_wire)
    assert sum(m.nbytes for m in msgs) == halo.recv_rows(1 + NT, 1, 0) * K * 8 > 250e6

    dt = 120.0 * (6.0 / 30.0) ** 2
    stream = oa.Stream()

    def two_steps(overlap, fuse_stages):
        state.copy_to_device(h, u, 0)
        upload()
        st = oa.TimeStepper("RungeKutta4", dt, tend, aux, mesh, halo, tracers)
        st.set_option("FuseStageUpdates", fuse_stages)
        st.set_option("OverlapHaloExchange", overlap)
        n_res = oa.device_resource_count()
        n_calls = len(calls)
        for _ in range(2):
            st.do_step(state, stream=stream)
        oa.device_synchronize()
        assert oa.device_resource_count() == n_res          # nothing is created inside a step
        assert len(calls) - n_calls == 4                     # two exchange points per RK4 step (RungeKutta4Stepper.cpp:95-131)
        h1, u1 = state.copy_to_host(0)
        tr1 = tracers.copy_to_host(0)
        del st
        return h1[:nc].copy(), u1[:ne].copy(), tr1[:, :nc].copy()

    ha, ua, ta = two_steps(True, True)
    assert np.isfinite(ha).all() and np.isfinite(ua).all() and np.isfinite(ta).all()    # EVERY owned value
    assert ha.min() > 0.5 and np.abs(ha - h[:nc]).max() > 0 and np.abs(ua - u[:ne]).max() > 0
    hb, ub, tb = two_steps(False, True)
    assert np.array_equal(ha, hb) and np.array_equal(ua, ub) and np.array_equal(ta, tb), "overlapped != sequential"
    del hb, ub, tb
    hc, uc, tc = two_steps(False, False)
    assert np.array_equal(ha, hc) and np.array_equal(ua, uc) and np.array_equal(ta, tc), "stage-fused != reference-structured"
    del hc, uc, tc
    # ---- ... and the oracle's two RK4 steps of the same rank (r6): orc_rk4_step with an exchange callback that does what
    # the frozen-halo wire does -- every halo row back to its initial state at both exchange points of a step
    # (RungeKutta4Stepper.cpp:107-113, 127-131) -- on the NT = 3 subset.  h, u and tracers 0 / 17 / 36 of the 37-tracer
    # GPU run must equal it bit for bit on owned elements.
    na, nea = mesh.NCellsAll, mesh.NEdgesAll

    def frozen_halo(hh, uu, tt):
        hh[nc:na] = h[nc:na]
        uu[ne:nea] = u[ne:nea]
        tt[:, nc:na] = tr3[:, nc:na]
    ost = orc.make_state(h, u, tr3)
    for _ in range(2):
        orc.step("rk4", ost, dt, exchange=frozen_halo)
    assert np.array_equal(ha, ost["h"][0][:nc]), "configs[4] rank: h after two RK4 steps differs from the oracle"
    assert np.array_equal(ua, ost["u"][0][:ne]), "configs[4] rank: u after two RK4 steps differs from the oracle"
    assert np.array_equal(ta[SUB], ost["tr"][0][:, :nc]), "configs[4] rank: tracers 0 / 17 / 36 after two RK4 steps differ"
