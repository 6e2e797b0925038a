"""One rank of a multi-rank test (launched by tests/test_multirank_*.py as a child process).

mode cpu: Decomp + Halo exchange lists from the product's HOST code (no device), the
          arithmetic by the CPU oracle, messages over gloo.  Checks (a) HaloTest-style
          exchange of global-ID arrays (reference test/base/HaloTest.cpp:41-100) and (b) two
          RK4 steps of the partitioned run against the single-rank oracle on owned elements.
mode gpu: the full product path on the GPU -- C++ Halo::exchange* with HIP pack/unpack kernels,
          RungeKutta4Stepper::doStep -- with the messages staged through gloo (both ranks may
          share one GPU), against the same single-rank oracle -- or, with --against-partitioned, against
          the PARTITIONED oracle of the same world size and halo width on every local element (what the
          reference itself prints at N ranks when the setting is not partition independent: its default
          HaloWidth 3 with the del4 terms on, RungeKutta4Stepper.cpp:107).
"""
import argparse
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["cpu", "gpu"], required=True)
    ap.add_argument("--rank", type=int, required=True)
    ap.add_argument("--world", type=int, required=True)
    ap.add_argument("--port", type=int, required=True)
    ap.add_argument("--nx", type=int, default=16)
    ap.add_argument("--ny", type=int, default=16)
    ap.add_argument("--levels", type=int, default=4)
    ap.add_argument("--tracers", type=int, default=2)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--stepper", default="RungeKutta4")
    ap.add_argument("--halo-width", type=int, default=3)
    ap.add_argument("--no-del4", action="store_true", help="disable the two radius-2 (del4) terms")
    ap.add_argument("--eddy-diff4", type=float, default=0.0,
                    help="EddyDiff4 (Default.yml: 0 with the term enabled): non-zero makes the tracers' radius-2 term count")
    ap.add_argument("--no-overlap", action="store_true", help="RK4 (gpu mode): exchange after the stage instead of overlapped")
    ap.add_argument("--user-stream", action="store_true",
                    help="gpu mode: step on a non-blocking stream created with omg_stream_create instead of the default stream")
    ap.add_argument("--local-order", default="global", choices=["global", "curve", "hilbert", "kd"],
                    help="Decomp local numbering: the reference's (global id) or along a Morton curve")
    ap.add_argument("--partition", default="rcb", choices=["rcb", "graph"], help="built-in partitioner (omg_partition_cells)")
    ap.add_argument("--rtol", type=float, default=0.0,
                    help="0 = owned elements must equal the single-rank run bit for bit; > 0 = the partitioned run may "
                         "deviate by at most this (relative to the field's max), and MUST deviate (the setting is known "
                         "not to be partition independent: HaloWidth 3 with the radius-2 del4 terms)")
    ap.add_argument("--wire", default="gloo", choices=["gloo", "ipc"],
                    help="gpu mode: gloo = host-staged test wire (synchronises the stream); ipc = the library's PeerWire "
                         "(HIP IPC mailboxes + flag kernels, fully stream-ordered: exercises the event ordering of the "
                         "overlapped exchanges for real)")
    ap.add_argument("--stress", type=int, default=0,
                    help="gpu mode: this many back-to-back exchanges of changing data, alternating between two non-blocking "
                         "streams, every one verified (flag / visibility races of a stream-ordered wire would show here)")
    ap.add_argument("--peer-timeout-test", action="store_true",
                    help="--wire ipc: rank 0 exchanges while the others never do: its wait kernel must give up after the "
                         "wire's time limit, raise the sticky status and make the next exchange fail loudly -- no hang")
    ap.add_argument("--against-partitioned", action="store_true",
                    help="gpu mode: the comparison is with the PARTITIONED oracle of the same world size and halo width "
                         "(this rank's oracle on this rank's local mesh, its exchanges over gloo at the reference's two "
                         "exchange points, RungeKutta4Stepper.cpp:107-113,127-131) -- what Omega itself prints at N ranks, "
                         "partition-dependent bits included -- on EVERY local element, owned and halo, bit for bit")
    ap.add_argument("--mesh", default="hex", help="hex (planar nx x ny) | any name of tests/meshes.py: icoN, fibN, "
                                                  "hexNXxNY, <base>_coast_<kind>[_raw][_compact], <base>_pad8")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(a.port), RANK=str(a.rank), WORLD_SIZE=str(a.world))
    dist.init_process_group("gloo", rank=a.rank, world_size=a.world)

    import omega_amd as oa
    from omega_amd.meshgen import planar_hex, synthetic_state
    from oracle import oracle as O
    from tests.problem import Problem
    from tests.meshes import named_mesh

    g = planar_hex(a.nx, a.ny, 30.0e3) if a.mesh == "hex" else named_mesh(a.mesh)
    K, NT, dt = a.levels, a.tracers, (5.0 if a.mesh.startswith("fib") else 600.0)   # (fib: a few very short edges)
    gpu = a.mode == "gpu"
    if gpu:
        oa.device_init(0)
    cfg = {"VelHyperDiffTendencyEnable": 0, "TracerHyperDiffTendencyEnable": 0} if a.no_del4 else {}
    if a.eddy_diff4:
        cfg["EddyDiff4"] = a.eddy_diff4
    P = Problem(g, K, NT, nparts=a.world, rank=a.rank, device=gpu, config=cfg, halo_width=a.halo_width,
                local_order=a.local_order, partition=a.partition)
    m = P.mesh
    halo = P.halo if gpu else oa.Halo(P.decomp)
    nbrs = halo.neighbors
    ids = {0: P.cell_id, 1: P.edge_id, 2: P.vertex_id}
    owned = {0: m.NCellsOwned, 1: m.NEdgesOwned, 2: m.NVerticesOwned}
    nall = {0: m.NCellsAll, 1: m.NEdgesAll, 2: m.NVerticesAll}
    sizes = {0: m.NCellsSize, 1: m.NEdgesSize, 2: m.NVerticesSize}

    # ---------------- host exchange over gloo using the product's lists (cpu mode) ----------------
    def host_exchange(arr, elem):
        """arr: [rows, K] or [NT, rows, K] numpy, in place."""
        a3 = arr if arr.ndim == 3 else arr[None]
        ops, recvs = [], []
        for i, t in enumerate(nbrs):
            sl, rl = halo.get_list(i, elem, False), halo.get_list(i, elem, True)
            sb = torch.from_numpy(np.ascontiguousarray(a3[:, sl, :]))
            rb = torch.empty((a3.shape[0], len(rl), a3.shape[2]), dtype=torch.float64)
            recvs.append((rl, rb))
            if rb.numel():
                ops.append(dist.P2POp(dist.irecv, rb, t))
            if sb.numel():
                ops.append(dist.P2POp(dist.isend, sb, t))
        for r in dist.batch_isend_irecv(ops) if ops else []:
            r.wait()
        for rl, rb in recvs:
            a3[:, rl, :] = rb.numpy()

    wire = None
    if gpu and a.wire == "ipc":
        rows = max(halo.recv_rows(3, 0, 0), halo.recv_rows(0, 3, 0), halo.recv_rows(0, 0, 3), halo.recv_rows(1 + NT, 1, 0))
        wire = oa.PeerWire(a.world, a.rank, max(rows, 1) * K * 8)
        handles = [None] * a.world
        dist.all_gather_object(handles, wire.handle())
        wire.connect(handles)
        halo.use_peer(wire)
    elif gpu:
        from tests.gloo_transport import GlooStagedTransport
        GlooStagedTransport(halo)

    if a.peer_timeout_test:
        assert wire is not None
        wire.set_timeout(1.0)
        if a.rank == 0:
            import time
            marked = np.zeros((sizes[0], K))
            marked[owned[0]: nall[0]] = -5.0
            buf = oa.DeviceBuffer(marked)
            t0 = time.time()
            halo.exchange(buf.ptr, 1, sizes[0], K, 0)        # the LAST exchange before a read-back: nothing follows it
            oa.device_synchronize()
            el = time.time() - t0
            assert 0.9 < el < 10.0, el                       # the wave left after the limit, not before, not never
            assert wire.info()["status"] == 2, wire.info()   # bit 1: the wait for the neighbours' messages gave up
            # the unpack kernel saw the raised status: the halo rows are what they were (not the mailbox's zeros) ...
            assert np.array_equal(buf.to_host().reshape(marked.shape), marked), "a failed exchange must not unpack a stale mailbox"
            try:                                             # ... and the host learns it without another exchange
                halo.check()
                raise AssertionError("omg_halo_check after a timed-out exchange must fail")
            except oa.OmegaAmdError as exc:
                assert "gave up" in str(exc), str(exc)
            try:
                halo.exchange(buf.ptr, 1, sizes[0], K, 0)
                raise AssertionError("an exchange after a timed-out one must fail")
            except oa.OmegaAmdError as exc:
                assert "timed out" in str(exc), str(exc)
        dist.barrier()
        wire.close()
        dist.destroy_process_group()
        print(f"rank {a.rank}/{a.world} OK (peer wire timeout)")
        return

    # ---------------- (a) HaloTest: global ids on owned, garbage on halo, exchange, compare ----------------
    for elem in (0, 1, 2):
        for nt in (1, 3):
            ref = np.zeros((nt, sizes[elem], K))
            for t in range(nt):
                ref[t, : nall[elem], :] = (ids[elem][: nall[elem], None] * 10.0 + t) + 0.001 * np.arange(K)[None, :]
            arr = ref.copy()
            arr[:, owned[elem]: nall[elem], :] = -999.0
            if gpu:     # (the library's own device buffers: torch never touches the GPU in this process)
                buf = oa.DeviceBuffer(arr)
                halo.exchange(buf.ptr, nt, sizes[elem], K, elem)
                oa.device_synchronize()
                arr = buf.to_host().reshape(ref.shape)
            else:
                host_exchange(arr if nt > 1 else arr[0], elem)
            assert np.array_equal(arr, ref), f"rank {a.rank}: halo exchange mismatch elem {elem} nt {nt}"

    # the reference's other element types (HaloTest.cpp:41-100 runs I4 / I8 / R4 / R8, rank 1-5): I4 rank 1 (the global ids
    # themselves), I4 rank 2 with an odd row length, R8 rank 1, I4 rank 3 -- through the same job-table kernels
    if gpu:
        for elem in (0, 1, 2):
            for dtype, nt, k in ((np.int32, 1, 1), (np.int32, 1, 3), (np.float64, 1, 1), (np.int32, 2, 5), (np.float32, 1, 2)):
                ref = np.zeros((nt, sizes[elem], k), dtype=dtype)
                for t in range(nt):
                    ref[t, : nall[elem], :] = ids[elem][: nall[elem], None] * 8 + t * 3 + np.arange(k)[None, :]
                arr = ref.copy()
                arr[:, owned[elem]: nall[elem], :] = -7
                buf = oa.DeviceBuffer(arr)
                halo.exchange(buf.ptr, nt, sizes[elem], k, elem, elem_bytes=arr.itemsize)
                oa.device_synchronize()
                got = buf.to_host()
                assert np.array_equal(got, ref), f"rank {a.rank}: halo exchange mismatch elem {elem} {dtype.__name__} nt {nt} k {k}"

    if gpu and a.stress > 0:
        streams = [oa.Stream(), oa.Stream()]
        elem, nt = 0, 2
        base = np.zeros((nt, sizes[elem], K))
        base[:, : nall[elem], :] = ids[elem][None, : nall[elem], None] * 3.0 + np.arange(K)[None, None, :]
        bufs = [oa.DeviceBuffer(base.copy()) for _ in range(7)]   # one per exchange of a batch: nothing is reused before
        for it in range(a.stress):                               # the batch has been synchronised and checked
            b, st = bufs[it % 7], streams[it % 2]
            ref = base + it
            ref[:, nall[elem]:, :] = 0.0
            arr = ref.copy()
            arr[:, owned[elem]: nall[elem], :] = -1.0          # halo rows must come from the owners, every time
            oa._chk(oa.lib().omg_copy_to_device(C.c_void_p(b.ptr), arr.ctypes.data_as(C.c_void_p), C.c_size_t(arr.nbytes)))
            halo.exchange(b.ptr, nt, sizes[elem], K, elem, stream=st)
            if it % 7 == 6 or it == a.stress - 1:               # seven exchanges queue up, then all of them are checked
                oa.device_synchronize()
                for j in range(it - it % 7, it + 1):
                    want = base + j
                    want[:, nall[elem]:, :] = 0.0
                    got = bufs[j % 7].to_host().reshape(want.shape)
                    assert np.array_equal(got, want), f"rank {a.rank}: stress exchange {j} wrong"

    # the steppers' exchange through the C ABI (omg_halo_exchange_state): h, u and the tracers as one message per
    # neighbour; the halo rows, overwritten here, must come back from their owners (the state is a function of the
    # global id, so the expected values are the rows uploaded at the start)
    if gpu:
        hp, up, tp = P.h.copy(), P.u.copy(), P.tr.copy()
        hp[owned[0]: nall[0]] = -7.0
        up[owned[1]: nall[1]] = -7.0
        tp[:, owned[0]: nall[0]] = -7.0
        P.state.copy_to_device(hp, up, 0)
        P.tracers.copy_to_device(tp, 0)
        halo.exchange_state(P.state, P.tracers if NT > 0 else None, 0)
        oa.device_synchronize()
        halo.check()
        hb, ub = P.state.copy_to_host(0)
        assert np.array_equal(hb[: nall[0]], P.h[: nall[0]]) and np.array_equal(ub[: nall[1]], P.u[: nall[1]])
        if NT > 0:
            assert np.array_equal(P.tracers.copy_to_host(0)[:NT, : nall[0]], P.tr[:NT, : nall[0]])

    # ---------------- (b) time stepping: partitioned run vs single-rank oracle ----------------
    okind = {"RungeKutta4": "rk4", "RungeKutta2": "rk2", "Forward-Backward": "fb"}[a.stepper]
    Mg = O.Mesh.single_rank(g, K)
    og = O.Oracle(Mg, NT, O.default_config(**cfg))
    hg, ug, trg = synthetic_state(g, K, NT)

    def pad(x):
        out = np.zeros(x.shape[:-2] + (x.shape[-2] + 1, x.shape[-1]))
        out[..., :-1, :] = x
        return out
    stg = og.make_state(pad(hg), pad(ug), pad(trg))
    for _ in range(a.steps):
        og.step(okind, stg, dt)

    def partitioned_oracle_run():
        stl = P.oracle.make_state(P.h, P.u, P.tr)

        def ex(hh, uu, tt):
            host_exchange(hh, 0)
            host_exchange(uu, 1)
            host_exchange(tt, 0)
        for _ in range(a.steps):
            P.oracle.step(okind, stl, dt, exchange=ex)
        return stl["h"][0], stl["u"][0], stl["tr"][0]

    if gpu and a.against_partitioned:
        ph, pu, ptr = partitioned_oracle_run()     # before the GPU run: the gloo group is idle, nothing else uses it yet
    if gpu:
        st = oa.TimeStepper(a.stepper, dt, P.tend, P.aux, P.mesh, halo, P.tracers)
        if a.stepper == "RungeKutta4":
            st.set_option("OverlapHaloExchange", not a.no_overlap)
            # the overlapped path splits the last kernels of a stage into a band and an interior launch:
            # make sure this mesh has both, or the test would not exercise it
            nb, ni = P.mesh.get_int("NBandCells"), P.mesh.get_int("NInteriorCells")
            assert nb > 0 and (ni > 0 or g["nCells"] <= 24 * 24), (nb, ni)
            # ... and the stages whose output is exchanged right away leave out the halo cells that finish nothing
            # owned here (MeshView::BandSendCells): fewer cells than the band, at least the owned band cells
            ns = P.mesh.get_int("NBandSendCells")
            assert nb - (m.NCellsAll - m.NCellsOwned) <= ns < nb, (ns, nb, m.NCellsOwned, m.NCellsAll)
        user_stream = oa.Stream() if a.user_stream else None   # hipStreamNonBlocking: no implicit ordering with stream 0
        # everything a step needs exists once the stepper does (RungeKutta4Stepper.cpp:43-64 allocates in finalizeInit):
        # no device buffer, stream or event is created inside doStep -- not even in the first one
        n_res = oa.device_resource_count()
        for _ in range(a.steps):
            st.do_step(P.state, stream=user_stream)
        oa.device_synchronize()
        assert oa.device_resource_count() == n_res, (n_res, oa.device_resource_count())
        h, u = P.state.copy_to_host(0)
        tr = P.tracers.copy_to_host(0)
    else:
        h, u, tr = partitioned_oracle_run()

    if gpu and a.against_partitioned:
        # the reference's own N-rank result (its default HaloWidth 3 with del4 on is not partition independent,
        # RungeKutta4Stepper.cpp:107: what has to be reproduced is the N-rank run, not the 1-rank one)
        na, nea = m.NCellsAll, m.NEdgesAll
        bad = []
        for name, got, want in (("h", h[:na], ph[:na]), ("u", u[:nea], pu[:nea]), ("tracers", tr[:NT, :na], ptr[:NT, :na])):
            if not np.array_equal(got, want):
                w = np.argwhere(got != want)
                bad.append(f"{name}: {len(w)} values differ, first at {w[0].tolist()} (owned cells {m.NCellsOwned}, edges "
                           f"{m.NEdgesOwned}), max |diff| {np.abs(got - want).max():.3e}")
        assert not bad, f"rank {a.rank}: GPU run differs from the partitioned oracle: " + "; ".join(bad)
        nc, ne = m.NCellsOwned, m.NEdgesOwned
        gh, gu = stg["h"][0][P.cell_id[:nc] - 1], stg["u"][0][P.edge_id[:ne] - 1]
        dev = max(np.abs(h[:nc] - gh).max() / np.abs(gh).max(), np.abs(u[:ne] - gu).max() / np.abs(gu).max())
        devs = [None] * a.world
        dist.all_gather_object(devs, float(dev))
        dist.barrier()
        if wire is not None:
            halo.check()
            wire.close()
        dist.destroy_process_group()
        print(f"rank {a.rank}/{a.world} OK ({a.mode}, {a.stepper}, {len(nbrs)} neighbours, equal to the partitioned oracle "
              f"on all {na} cells / {nea} edges; deviation from the 1-rank run {max(devs):.3e})")
        return

    nc, ne = m.NCellsOwned, m.NEdgesOwned
    gh = stg["h"][0][P.cell_id[:nc] - 1]
    gu = stg["u"][0][P.edge_id[:ne] - 1]
    gtr = stg["tr"][0][:, P.cell_id[:nc] - 1]
    assert np.isfinite(gh).all() and np.isfinite(gu).all()
    if a.rtol > 0:
        parts = (np.abs(h[:nc] - gh).max() / np.abs(gh).max(), np.abs(u[:ne] - gu).max() / np.abs(gu).max(),
                 (np.abs(tr[:NT, :nc] - gtr[:NT]).max() / np.abs(gtr[:NT]).max()) if NT else 0.0)
        print(f"rank {a.rank}: relative deviation h {parts[0]:.3e}  u {parts[1]:.3e}  tracers {parts[2]:.3e}")
        dev = max(parts)
        devs = [None] * a.world
        dist.all_gather_object(devs, float(dev))
        assert max(devs) <= a.rtol, f"rank {a.rank}: deviation {max(devs):.3e} from the single-rank run exceeds {a.rtol:.1e}"
        assert max(devs) > 0.0, "expected a (small) partition dependence in this setting"
        dist.barrier()
        dist.destroy_process_group()
        print(f"rank {a.rank}/{a.world} OK ({a.mode}, {a.stepper}, {len(nbrs)} neighbours, max deviation {max(devs):.3e})")
        return
    assert np.array_equal(h[:nc], gh), f"rank {a.rank}: h differs from the single-rank run (max {np.abs(h[:nc]-gh).max()})"
    assert np.array_equal(u[:ne], gu), f"rank {a.rank}: u differs from the single-rank run (max {np.abs(u[:ne]-gu).max()})"
    assert np.array_equal(tr[:NT, :nc], gtr[:NT]), f"rank {a.rank}: tracers differ from the single-rank run"
    # the halo after the end-of-step exchange: the owners' values, on every layer
    na, nea = m.NCellsAll, m.NEdgesAll
    assert np.array_equal(h[nc:na], stg["h"][0][P.cell_id[nc:na] - 1]), f"rank {a.rank}: halo h differs after the step"
    assert np.array_equal(u[ne:nea], stg["u"][0][P.edge_id[ne:nea] - 1]), f"rank {a.rank}: halo u differs after the step"
    assert np.array_equal(tr[:NT, nc:na], stg["tr"][0][:NT, P.cell_id[nc:na] - 1]), f"rank {a.rank}: halo tracers differ"
    note = ""
    if wire is not None:
        # globalSum inside the library (omg_halo_global_sum_dd: the (hi, lo) pairs travel through the wire's gather slots
        # and are combined in rank order): the N-rank sums of h, u and every tracer over owned elements are the one-rank
        # sums of the single-rank oracle's state, bit for bit, and equal on every rank
        import math
        kp = oa.level_pitch(K)
        ones_c, ones_e = oa.DeviceBuffer(np.ones(sizes[0])), oa.DeviceBuffer(np.ones(sizes[1]))
        pairs = [oa.local_weighted_sum_dd(ones_c.ptr, P.state.device_ptr(0, 0), nc, K, row_pitch=kp),
                 oa.local_weighted_sum_dd(ones_e.ptr, P.state.device_ptr(1, 0), ne, K, row_pitch=kp)]
        pairs += [oa.local_weighted_sum_dd(ones_c.ptr, P.tracers.device_ptr(0) + 8 * l * sizes[0] * kp, nc, K, row_pitch=kp)
                  for l in range(NT)]
        sums = halo.global_sum_dd(pairs)
        want = [math.fsum(stg["h"][0][:-1].ravel()), math.fsum(stg["u"][0][:-1].ravel())]
        want += [math.fsum(stg["tr"][0][l, :-1].ravel()) for l in range(NT)]
        assert sums == want, f"rank {a.rank}: global sums {sums} != one-rank sums {want}"
        every = [None] * a.world
        dist.all_gather_object(every, sums)
        assert all(e == sums for e in every), "the ranks disagree on the global sums"
        assert oa.global_sum_dd(pairs[0], halo=halo) == want[0]
        halo.check()        # the host is synchronised with every exchange: the wire must report none as failed
        info = wire.info()
        assert info["status"] == 0 and info["exchanges"] >= 6 + 15 + a.steps + a.stress, info
        note = f", peer wire: {info['exchanges']} exchanges"
    dist.barrier()          # every rank's GPU work is complete (device_synchronize above): mailboxes may go
    if wire is not None:
        wire.close()
    dist.destroy_process_group()
    print(f"rank {a.rank}/{a.world} OK ({a.mode}, {a.stepper}, {len(nbrs)} neighbours{note})")


if __name__ == "__main__":
    main()
