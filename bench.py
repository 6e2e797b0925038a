#!/usr/bin/env python
"""bench.py -- headline benchmark of the Omega ocean-dycore hot path on MI355X.

Metric (BASELINE.json): tendency cell-level-updates/s of the fused RHS
(Tendencies::computeAllTendencies) and SYPD of the RK4 step, on a synthetic QU30-sized
mesh (planar periodic 680 x 680 = 462,400 cells ~ QU30's 460k), 80 levels, 6 tracers,
Default.yml term set (del2 + del4, center fluxes).  One process per GPU; for N > 1 the mesh
is partitioned N ways (strong scaling) and halos travel as RCCL send/recv over xGMI.

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one full RHS evaluation over the whole mesh (all N ranks).  The JSON line also
carries `sypd` (RK4 steps, halo exchanges included), `roofline` (dominant kernel, live HIP
events on the launch stream) and, at N = 1, `cpu_baseline` (the CPU oracle timed on a
the same full mesh -- a bounded number of evaluations -- on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

# dmabuf IPC (the pool's hosts support no other): RCCL's and the peer wire's cross-process memory handles need it; the
# image exports it, a launcher with a scrubbed environment may not -- set before anything loads the HIP runtime
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

oa = None   # omega_amd: loaded by load_library() in a process that computes, never by the launcher parent


def load_library():
    """libomega_amd.so (and with it ROCm's own libamdhip64 / librccl) before anything imports torch.  Called by the rank
    processes only: the parent of a self-launched N > 1 run (launch_ranks) never loads the HIP runtime."""
    global oa, icosahedral_points, planar_hex, reorder_cells_blocked, reorder_cells_morton
    global spherical_voronoi, synthetic_state, synthetic_state_rows
    import omega_amd
    omega_amd.lib()
    from omega_amd import meshgen
    oa = omega_amd
    icosahedral_points, planar_hex = meshgen.icosahedral_points, meshgen.planar_hex
    reorder_cells_blocked, reorder_cells_morton = meshgen.reorder_cells_blocked, meshgen.reorder_cells_morton
    spherical_voronoi, synthetic_state, synthetic_state_rows = (meshgen.spherical_voronoi, meshgen.synthetic_state,
                                                                meshgen.synthetic_state_rows)


HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)

WORKLOADS = {
    # name: (nx, ny, dc [m], levels, tracers, description)
    "qu30": (680, 680, 30.0e3, 80, 6, "QU30-sized planar periodic hex mesh 680x680 (462400 cells), 80L, 6 tracers"),
    "qu240": (84, 84, 240.0e3, 60, 2, "QU240-sized planar periodic hex mesh 84x84 (7056 cells), 60L, 2 tracers"),
    "ec30to60": (484, 484, 45.0e3, 60, 2, "EC30to60-sized planar mesh 484x484 (234256 cells), 60L, 2 tracers"),
    "qu30_eighth": (340, 170, 30.0e3, 80, 6, "one eighth of the QU30-sized mesh (340x170 = 57800 cells), 80L, 6 tracers"),
    "qu30_quarter": (340, 340, 30.0e3, 80, 6, "one quarter of the QU30-sized mesh (340x340 = 115600 cells), 80L, 6 tracers"),
    "orrs18to6_eighth": (680, 680, 6.0e3, 80, 37,
                         "one eighth of an oRRS18to6-sized mesh (680x680 = 462400 of 3.7M cells), 80L, 37 tracers"),
    "ico7": (0, 7, 0.0, 80, 6, "spherical icosahedral Voronoi mesh, 163842 cells (12 pentagons), 80L, 6 tracers"),
    "ico8": (0, 8, 0.0, 80, 6, "spherical icosahedral Voronoi mesh, 655362 cells (12 pentagons), 80L, 6 tracers "
                               "(QU30-sized on the sphere; the mesh generator needs a few minutes)"),
    "fib7": (0, -163842, 0.0, 80, 6, "spherical Voronoi mesh of a relaxed Fibonacci lattice, 163842 cells (pentagons, "
                                     "hexagons AND heptagons: maxEdges 7 with valence 6 dominant, as real MPAS meshes), 80L, 6 tracers"),
    "ico6": (0, 6, 0.0, 60, 2, "spherical icosahedral Voronoi mesh, 40962 cells (12 pentagons), 60L, 2 tracers"),
    "ico5": (0, 5, 0.0, 60, 2, "QU240-sized ON THE SPHERE: icosahedral Voronoi mesh, 10242 cells (12 pentagons), 60L, 2 tracers"),
    "small": (96, 96, 30.0e3, 80, 6, "small smoke workload 96x96, 80L, 6 tracers"),
    "orrs18to6": (1924, 1924, 6.0e3, 80, 37,
                  "oRRS18to6-sized planar periodic hex mesh 1924x1924 (3701776 cells), 80L, 37 tracers -- BASELINE configs[4]; "
                  "needs 8 GPUs (59 GB of arrays per rank)"),
    "hex405": (404, 406, 49.0e3, 80, 6, "planar periodic hex mesh of the size of ico7 / fib7 (404x406 = 164024 cells; the "
                                        "periodic generator needs an even row count), 80L, 6 tracers: the planar reference of the "
                                        "spherical workloads"),
    # culled meshes (land removed the way MPAS ocean meshes are: omega_amd/meshgen.py cull / coast_mask "continents")
    "qu30_coast": (800, 800, 30.0e3, 80, 6, "QU30-sized CULLED planar mesh: 800x800 hexagons with 28 % land removed "
                                            "(continents + one-cell islands), 80L, 6 tracers"),
    "ico7_coast": (0, 7, 0.0, 80, 6, "spherical icosahedral Voronoi mesh of 163842 cells with 28 % land removed, 80L, 6 tracers"),
    "fib7_coast": (0, -163842, 0.0, 80, 6, "relaxed Fibonacci sphere (valences 5, 6, 7) of 163842 cells with 28 % land removed, "
                                           "80L, 6 tracers"),
    "small_coast": (96, 96, 30.0e3, 80, 6, "small smoke workload 96x96 with 28 % land removed, 80L, 6 tracers"),
    # a mesh file whose per-cell lists are not in ring order for 1 % of the cells (meshgen.permute_cell_slots): those cells
    # run through the generic per-cell bodies (MeshView::BadCells), everything else keeps the fast kernels
    "hex405_perm1": (404, 406, 49.0e3, 80, 6, "the hex405 mesh with two slots of edgesOnCell / cellsOnCell / verticesOnCell "
                                             "swapped in 1 % of the cells, 80L, 6 tracers"),
}


def cached_mesh(key, build):
    """Spherical meshes take 45-75 s to generate (scipy Voronoi + Lloyd sweeps): a profiling session that runs bench.py
    ten times over generates each once and keeps it as an .npz under $OMEGA_MESH_CACHE (default /tmp/omega_amd_mesh_cache;
    the cache never travels: a fresh box generates afresh)."""
    d = os.environ.get("OMEGA_MESH_CACHE", "/tmp/omega_amd_mesh_cache")
    f = os.path.join(d, key + ".npz")
    if os.path.exists(f):
        with np.load(f) as z:
            return {k: (z[k] if z[k].ndim else z[k].item()) for k in z.files}
    g = build()
    try:
        os.makedirs(d, exist_ok=True)
        tmp = f + f".{os.getpid()}.tmp.npz"
        np.savez(tmp, **g)
        os.replace(tmp, f)
    except OSError:
        pass
    return g


def workload_mesh(workload, block=1, max_edges=0):
    """The global mesh of a named workload, as a mesh file would hold it (load_library() first).  Also what the parity
    tests at bench size build their meshes with (tests/test_gpu_parity.py)."""
    nx, ny, dc = WORKLOADS[workload][:3]
    if workload.startswith("fib"):
        g = cached_mesh(f"fib_{-ny}_l2", lambda: spherical_voronoi(-ny, lloyd=2))
    elif workload.startswith("ico"):   # sphere: cells already numbered along a Morton curve in (lon, z)
        g = cached_mesh(f"ico_{ny}", lambda: spherical_voronoi(points=icosahedral_points(ny), lloyd=0))
    else:
        g = planar_hex(nx, ny, dc)
    if workload.startswith(("ico", "fib")):
        pass
    elif block <= 0:
        g = reorder_cells_morton(g, hilbert=block < 0)
    elif block > 1:
        g = reorder_cells_blocked(g, block)
    if workload.endswith("_coast"):
        from omega_amd.meshgen import coast_mask, cull
        g = cull(g, coast_mask(g, "continents"))
    if workload.endswith("_perm1"):
        from omega_amd.meshgen import permute_cell_slots
        g = permute_cell_slots(g, 0.01)
    if max_edges > g["maxEdges"]:
        from omega_amd.meshgen import pad_max_edges
        g = pad_max_edges(g, max_edges)
    return g


def algorithmic_bytes_per_cell_level(nt, kernel=None):
    """SURVEY.md 8(d) B_staged = 8*(39 + 5*NT) B per cell-level for the whole RHS (NE = 3NC,
    NV = 2NC); per kernel: the arrays that kernel must read / write once."""
    per_kernel = {
        "VortVertexBody": 8 * (1 + 3 + 2 * 2),                 # h, u -> RelVort, 1/LayerThickVertex
        "FusedCell1Body": 8 * (3 + 1 + nt + 3 + nt),            # u, h, tr -> KE, Div, hTend, Del2Tr
        # vertex pass + side-0 PV sums folded in: + RelVort, 1/LayerThickVertex (2 vertex arrays) and the PV sums
        "FusedCellL1PVBody": 8 * (3 + 1 + nt + 3 + nt + 2 * 2 + 3),
        "Del2CellRingBody": 8 * (1 + 2 + 1),                    # Div, RelVort -> Del2Div
        "Del2VertexSelBody": 8 * (1 + 2 + 2),                   # Div, RelVort -> Del2RelVort
        "CellPVBody<side 0>": 8 * (3 + 1 + 2 * 2 + 3),          # u, h, RelVort, InvThick -> running PV sums
        "CellPVFinalBody": 8 * (3 + 1 + 2 * 2 + 3 + 3 + 2 + 3),  # + sums, KE, Div, Del2Div, Del2RV -> uTend
        "FusedCell3Body": 8 * (nt + nt + 1 + 3 + nt),           # tr, Del2Tr, h, u -> trTend
    }
    # independent sweeps sharing one launch (KernelCommon.h: tileKernel2)
    per_kernel["Del2CellRingBody+Del2VertexSelBody"] = per_kernel["Del2CellRingBody"] + per_kernel["Del2VertexSelBody"]
    per_kernel["CellPVFinalBody+FusedCell3Body"] = per_kernel["CellPVFinalBody"] + per_kernel["FusedCell3Body"]
    # both in one thread: h and u gathered once
    per_kernel["CellPVFinalTracerBody"] = per_kernel["CellPVFinalBody"] + per_kernel["FusedCell3Body"] - 8 * (1 + 3)
    per_kernel["CellPVFinalBody (rarer valences)"] = per_kernel["CellPVFinalBody"]
    per_kernel["CellPVBody<side 0>+FusedCell3Body"] = 8 * (nt + nt + 1 + 3 + nt + 2 * 2 + 3)
    per_kernel["FusedDel2CellBody"] = per_kernel["Del2CellRingBody"]
    per_kernel["FusedDel2VertexBody"] = per_kernel["Del2VertexSelBody"]
    per_kernel["CellPVBody<side 1>+EdgeFinalBody"] = per_kernel["CellPVFinalBody"] + 8 * (3 + 3 + 1)
    for k in ("FusedEdgeChainBody", "FusedEdgeBody", "edgePatchKernel"):
        per_kernel[k] = 8 * (3 + 1 + 3 * 2 + 3 + 2 + 3)         # u,h,3V,KE,Div,Del2Div,Del2RV -> uTend
    if kernel is None:
        return 8 * (39 + 5 * nt)
    return per_kernel[kernel]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("OMEGA_BENCH_WORKLOAD", "qu30"), choices=sorted(WORKLOADS))
    ap.add_argument("--rk4-steps", type=int, default=-1, help="RK4 steps for SYPD (default: max(2, steps//4))")
    ap.add_argument("--dt", type=float, default=0.0,
                    help="time step [s]; 0 = Default.yml's 10 minutes at 30 km cells, scaled with the workload's cell size")
    ap.add_argument("--block", type=int, default=1,
                    help="cell numbering of the synthetic INPUT mesh (what a mesh file would hold): 1 = row-major (default), "
                         "0 = Morton curve, -1 = Hilbert curve, n > 1 = n x n blocks")
    ap.add_argument("--partition", default="graph", choices=["graph", "rcb"],
                    help="N > 1: built-in partitioner of the cell graph (graph = recursive graph bisection + FM / k-way "
                         "refinement, the stand-in for the reference's METIS k-way; rcb = coordinate bisection)")
    ap.add_argument("--max-edges", type=int, default=0,
                    help="store the mesh with this (larger) maxEdges dimension, as mesh files often do (padding slots)")
    ap.add_argument("--local-order", default="kd", choices=["curve", "hilbert", "kd", "global"],
                    help="local numbering chosen by Decomp (the library owns data locality): kd = k-d order, every aligned "
                         "run of 8 / 16 / 32 local cells a compact patch on the surface (default; against the Morton curve: "
                         "RHS -1..-3 %% planar, -2.8 %% on the icosahedral sphere), curve / hilbert = along a Morton / Hilbert "
                         "curve through the cell centres, global = the reference's global-id order")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", action="store_true",
                    help="N = 1: skip the two rocprofv3 --pmc child runs (FETCH_SIZE; WRITE_SIZE) that measure roofline.traffic "
                         "in this very run; traffic then comes from a committed profiles/*_pmc.json measured on the same kernel "
                         "sources, or is null")
    ap.add_argument("--realistic", default="auto",
                    help="N = 1: after the headline measurement, the same W + K evaluations on a realistically shaped mesh "
                         "(record key `realistic`).  auto (default) = fib7_coast when the headline workload is qu30, none "
                         "otherwise; `none`; or a workload name")
    ap.add_argument("--unfused", action="store_true", help="time the reference-structured launch sequence instead")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: exchange halos after the producing stage instead of overlapped with its interior part")
    ap.add_argument("--no-fuse-stages", action="store_true",
                    help="RK4 with the separate update kernels instead of stage updates folded into the RHS kernels")
    ap.add_argument("--backend", default="auto", choices=["auto", "rccl", "nccl", "peer", "gloo"],
                    help="halo wire for N > 1.  rccl (= nccl): RCCL send/recv over xGMI issued inside the library; STRICT -- "
                         "if the communicator cannot be created the RHS record is printed with rk4.error set and the run "
                         "exits non-zero.  peer: the library's direct peer-copy wire (HIP IPC mailboxes + flag kernels, "
                         "PeerWire.h), equally stream-ordered and device-to-device.  auto (default): rccl, and if RCCL cannot "
                         "be initialised on every rank, peer -- never a host-staged wire; config.halo_wire names what ran and "
                         "why.  gloo: host-staged rehearsal of the N>1 code path (needs --allow-host-staged)")
    ap.add_argument("--allow-host-staged", action="store_true",
                    help="permit --backend gloo: messages staged through host memory (a correctness rig, not a measurement "
                         "of the halo wire)")
    ap.add_argument("--settle-ms", type=float, default=60.0,
                    help="untimed: before the W warm-up steps (and before the RK4 warm-up step) the GPU runs this workload's "
                         "own RHS for this long, so that a short timed region does not start inside the power-management "
                         "transient that follows a load step (tools/probes/step_ramp.py); 0 = off")
    ap.add_argument("--rk4-timeout", type=float, default=240.0,
                    help="N > 1: seconds the stepping part (wire set-up, RK4 steps, cross-check) may take before rank 0 prints "
                         "the record with rk4.error set and the run ends non-zero; 0 = wait for ever")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--halo-width", type=int, default=0,
                    help="0 = 4 for N > 1 (partition-independent results with the del4 terms: two RHS evaluations per "
                         "exchange consume 2 x 2 layers), 3 for N = 1; the reference default is 3")
    ap.add_argument("--launch-timeout", type=float, default=1500.0,
                    help="self-launched N > 1 run (plain `python bench.py --gpus N`): seconds the rank processes may take "
                         "altogether before the parent ends them; once one rank has left, the others get --rk4-timeout + 60 s")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset): this process becomes the
    launcher.  It starts N FRESH child processes of this same script -- RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT set the way torch.distributed.run sets them, one rank per GPU --, passes rank 0's stdout (the ONE JSON
    line) through, sends everybody's stderr to its own, and returns non-zero if any rank does.  The parent never loads
    libomega_amd / the HIP runtime and never execs: every rank is a child created by subprocess.Popen before anything
    in this process has touched a GPU.  Children are ended by their exact PIDs only (a rank that outlives the first
    leaver by --rk4-timeout + 60 s, or the whole run --launch-timeout)."""
    import socket
    import subprocess
    import threading
    N = args.gpus
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(N):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(N), LOCAL_WORLD_SIZE=str(N),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMEGA_BENCH_LAUNCHER="self")
        env.setdefault("OMP_NUM_THREADS", "1")      # what torch.distributed.run does for its workers
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *argv], env=env, cwd=os.getcwd(),
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    lines = []

    def relay():        # rank 0's stdout: the record, relayed line by line as it comes
        for raw in procs[0].stdout:
            line = raw.decode(errors="replace")
            lines.append(line)
            sys.stdout.write(line)
            sys.stdout.flush()
    th = threading.Thread(target=relay, daemon=True)
    th.start()
    t0 = time.time()
    first_exit = None
    killed = []
    while any(p.poll() is None for p in procs):
        now = time.time()
        codes = [p.poll() for p in procs]
        if first_exit is None and any(c is not None for c in codes):
            first_exit = now
        grace = 20.0 if any(c not in (None, 0) for c in codes) else (args.rk4_timeout + 60.0)
        if now - t0 > args.launch_timeout or (first_exit is not None and now - first_exit > grace):
            for r, p in enumerate(procs):
                if p.poll() is None:
                    killed.append(r)
                    p.terminate()
            t1 = time.time()
            while any(p.poll() is None for p in procs) and time.time() - t1 < 10.0:
                time.sleep(0.1)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.05)
    for p in procs:
        p.wait()
    th.join(timeout=10.0)
    codes = [p.returncode for p in procs]
    if killed:
        print(f"[bench launcher] ended rank(s) {killed}: exit codes {codes} (rendezvous 127.0.0.1:{port})", file=sys.stderr, flush=True)
    elif any(codes):
        # (the port was free when probed a moment before the ranks started; a rendezvous failure right at the start of
        # rank 0's log means another process took it in between: run again)
        print(f"[bench launcher] rank exit codes {codes} (rendezvous 127.0.0.1:{port})", file=sys.stderr, flush=True)
    if not lines:
        print("[bench launcher] rank 0 printed no record", file=sys.stderr, flush=True)
        return 5
    bad = [c for c in codes if c]
    if not bad:
        return 0
    positive = [c for c in bad if c > 0]     # (a negative code = ended by a signal)
    return positive[0] if positive else 1


def main():
    args = parse_args()
    world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if world == 0 and args.gpus > 1:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    world = max(world, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE = {world} ranks")
    N = world
    live = live_note = None
    if N == 1 and not args.no_live_traffic:
        live, live_note = live_traffic(args)      # (child processes; this one has not loaded the HIP runtime yet)
        if live is None:
            print(f"[bench] no live HBM-traffic measurement: {live_note}", file=sys.stderr, flush=True)
    load_library()
    launcher = ("bench.py itself (fresh child processes per rank)" if os.environ.get("OMEGA_BENCH_LAUNCHER") == "self" else
                "torch.distributed.run (or an equivalent that sets WORLD_SIZE / RANK)" if "WORLD_SIZE" in os.environ else
                "none (one process)")

    nx, ny, dc, K, NT, desc = WORKLOADS[args.workload]
    if args.workload == "orrs18to6" and N < 8:
        raise SystemExit("--workload orrs18to6 is BASELINE configs[4] (3.7 M cells x 80 levels x 37 tracers: ~470 GB of arrays): "
                         "run it with --gpus 8; its per-GPU share on one GPU is --workload orrs18to6_eighth")
    if args.dt <= 0:
        ncell_sphere = -ny if ny < 0 else 10 * 4 ** ny + 2
        cell = dc if dc > 0 else (4.0 * np.pi * 6371.22e3 ** 2 / ncell_sphere) ** 0.5   # sphere: mean spacing
        # (below 30 km the fixed Default.yml del4 viscosity limits the step like cell^4: use cell^2 as a compromise)
        args.dt = 600.0 * cell / 30.0e3 if cell >= 30.0e3 else 600.0 * (cell / 30.0e3) ** 2
    dist = None
    stream = None
    if args.backend == "nccl":
        args.backend = "rccl"
    # stdout carries exactly ONE line, the JSON record: libraries that chat on stdout (gloo prints its connection
    # summary there) are sent to stderr until the record is printed
    sys.stdout.flush()
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    if N > 1:
        # torch.distributed is the rendezvous / side channel only (gloo, CPU): it distributes the RCCL unique id and
        # carries the timing reductions; the halo data path is RCCL inside libomega_amd.  torch never touches the GPU.
        import datetime
        import torch
        import torch.distributed as dist
        if args.single_device:
            local_rank = 0
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")   # one node: the side channel stays on the loopback device
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=300))
    if N == 1 and rank == 0 and not args.no_cpu_baseline:
        from oracle import oracle as O   # the checker's -O3 -march=native build: compiled (a child `make`) before this
        O.use_native_build()             # process touches the GPU
    oa.device_init(local_rank)
    halo_width = args.halo_width if args.halo_width > 0 else (4 if N > 1 else 3)

    t0 = time.time()
    g = workload_mesh(args.workload, block=args.block, max_edges=args.max_edges)
    gm = oa.GlobalMesh(g)
    cell_task, edge_cut = (oa.partition_cells(gm, N, args.partition) if N > 1 else (None, 0))
    decomp = oa.Decomp(gm, N, rank, halo_width, cell_task=cell_task, local_order=args.local_order)
    mesh = oa.HorzMesh(decomp, K)
    halo = oa.Halo(decomp) if N > 1 else None
    cell_id, edge_id = decomp.get_array("CellID"), decomp.get_array("EdgeID")
    # The state is synthesised PER RANK: the rows of this rank's local elements only, each value a function of (global
    # id, level) alone (meshgen.synthetic_state_rows), so that no rank ever holds a global [nCells, K] array -- 88 GB for
    # configs[4] -- and an N = 1 and an N = 8 run still start from the same bits (rk4.state_checksums_after_2_steps is the
    # check).  The reference initialises per task too (components/omega/src/ocn/OceanState.cpp:65-117).
    cells0, edges0 = cell_id[: mesh.NCellsAll] - 1, edge_id[: mesh.NEdgesAll] - 1
    kp = oa.level_pitch(K)

    def local_rows(a2d, rows):      # [n, K] -> [rows, K] with the zero sentinel row(s) behind
        out = np.zeros((rows, a2d.shape[1]))
        out[: a2d.shape[0]] = a2d
        return out

    def make_hu():
        hh, uu, _ = synthetic_state_rows(g, K, 0, cells0, edges0, tracers=[])
        if "boundaryEdge" in g:     # an ocean state: no normal flow through the coast
            uu[np.asarray(g["boundaryEdge"])[edges0] != 0] = 0.0
        return local_rows(hh, mesh.NCellsSize), local_rows(uu, mesh.NEdgesSize)

    def upload_tracers(trc):
        """one tracer at a time (a rank's 37 tracers of configs[4] are 11 GB: never as one host array)"""
        base = trc.device_ptr(0)
        for l in range(NT):
            t2 = synthetic_state_rows(g, K, NT, cells0, edges0[:0], tracers=[l])[2][0]
            buf = np.zeros((mesh.NCellsSize, kp))
            buf[: t2.shape[0], :K] = t2
            oa.copy_to_device(base + 8 * l * mesh.NCellsSize * kp, buf)

    h, u = make_hu()

    comm = None
    wire_note = None
    stream = oa.Stream()
    if N > 1 and args.backend == "gloo" and not args.allow_host_staged:
        raise SystemExit("--backend gloo stages every halo message through host memory; pass --allow-host-staged to "
                         "run it as a rehearsal of the N > 1 code path")

    cfg = oa.default_config()
    state = oa.OceanState(mesh, halo, K, 2)
    tracers = oa.Tracers(mesh, halo, K, NT, 2)
    aux = oa.AuxiliaryState(mesh, halo, K, NT)
    tend = oa.Tendencies(mesh, K, NT, cfg)
    tend.set_fused(not args.unfused)
    state.copy_to_device(h, u, 0)
    upload_tracers(tracers)
    setup_s = time.time() - t0

    def barrier():
        oa.device_synchronize()
        if N > 1:
            dist.barrier()
            oa.device_synchronize()

    def allmax(x):
        if N == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def settle():
        """Untimed.  A GPU that goes from idle to this load runs its first ~ 25 ms of kernels up to 12 % slower (a hump in
        the per-evaluation times, not a ramp: tools/probes/step_ramp.py, profiles/r03_probe_step_ramp.json); a timed
        region of K x 0.7 ms -- one rank of an 8-way run -- would sit inside it, one of K x 5 ms would not notice.  So
        the device is kept under the workload's own load for --settle-ms before the W warm-up steps.  Nothing of the
        timed region changes: still W untimed steps, then exactly K timed ones between the brackets."""
        n, t0 = 0, time.perf_counter()
        while args.settle_ms > 0 and (time.perf_counter() - t0) * 1e3 < args.settle_ms:
            for _ in range(4):
                tend.compute_all_tendencies(state, aux, tracers, stream=stream)
            oa.device_synchronize()
            n += 4
        return n

    # ------------------------------------------------ RHS: W warm-up + K timed steps
    settle_evals = settle()
    for _ in range(args.warmup):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    barrier()
    ev0, ev1 = oa.Event(), oa.Event()
    t_start = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    ev1.record(stream)
    # closing bracket: this rank's K steps are done (device synchronised) -> its elapsed time; then the barrier; the
    # job's time is the MAX over ranks (allmax below).  The barrier's own latency (a gloo round over the side channel)
    # is not part of anybody's K steps.
    oa.device_synchronize()
    wall_rhs = time.perf_counter() - t_start
    barrier()
    dev_ms = ev0.elapsed_ms(ev1)
    graph_stats = tend.graph_stats()
    # per-kernel durations for the roofline: HIP events between the launches, in a second pass of the same K steps
    # directly after the timed region (which therefore carries no event records between its kernels)
    tend.kernel_timing(True)
    for _ in range(args.steps):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    oa.device_synchronize()
    tend.kernel_timing(False)
    ktimes = tend.collect_kernel_times()
    wall_rhs = allmax(wall_rhs)
    ms_per_step = 1e3 * wall_rhs / args.steps

    n_cells_global = g["nCells"]
    cell_levels = n_cells_global * K
    value = cell_levels / (wall_rhs / args.steps)

    # ------------------------------------------------ roofline of the dominant kernel (rank 0's view)
    roofline = None
    if ktimes:
        name, ms = max(ktimes, key=lambda kv: kv[1])
        local_cell_levels = mesh.NCellsAll * K  # every launch sweeps owned + halo elements
        ach = algorithmic_bytes_per_cell_level(NT, name) * local_cell_levels / (ms * 1e-3) / 1e9
        # RHS-level figure: B_staged over the device time of one evaluation in the TIMED region (HIP events around it on
        # the launch stream); the sum of the per-kernel events of the second pass is reported next to it
        rhs_ms = dev_ms / args.steps
        kernels_sum_ms = sum(m for _, m in ktimes)
        rhs_ach = algorithmic_bytes_per_cell_level(NT) * local_cell_levels / (rhs_ms * 1e-3) / 1e9
        # HBM bytes per launch of that kernel from the committed PMC passes (tools/profile_bench.sh: rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE in separate runs of this same command, gfx950 correction applied).  Only a file
        # measured on exactly the kernel sources that run here (hash recorded in the file) and on this workload
        # counts; otherwise traffic is null rather than stale.
        traffic = traffic_src = None

        def bases(n):     # "A<6, 6>+B<7, 7>" -> "A+B" (template arguments contain commas and, nested, '+' never)
            # (CellPVFinalTracerPatchBody is CellPVFinalTracerBody with its tracer loop through LDS patches: the
            # library reports both under the latter name)
            return "+".join(x.split("<")[0].strip().replace("TracerPatchBody", "TracerBody") for x in n.split("+"))
        if live:
            cands = sorted((v for k, v in live.items() if bases(k) == bases(name)))
            if cands:
                traffic, traffic_src = cands[0], live_note
        if traffic is None and N == 1 and not args.unfused:
            import glob
            from tools.summarise_profile import kernel_source_sha
            sha = kernel_source_sha()
            for pmc_file in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc.json")), reverse=True):
                with open(pmc_file) as fh:
                    pmc = json.load(fh)
                wl = "qu30"
                pa = pmc.get("bench_args") or []
                if "--workload" in pa:
                    wl = pa[pa.index("--workload") + 1]
                if pmc.get("kernel_source_sha") != sha or wl != args.workload:
                    continue
                base = bases(name)
                cands = [k for k, rec in pmc.items() if isinstance(rec, dict) and bases(k) == base]
                # the RK4 stage-fused instantiations of the same body move more bytes (accumulator, provisional
                # state); the RHS timed here is the plain instantiation: the candidate with the fewest bytes
                plain = sorted(cands, key=lambda k: pmc[k]["hbm_bytes_per_launch"])
                if plain:
                    traffic = pmc[plain[0]]["hbm_bytes_per_launch"]
                    traffic_src = os.path.relpath(pmc_file, ROOT) + " (kernel_source_sha " + sha + ")"
                    break
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": algorithmic_bytes_per_cell_level(NT, name) * local_cell_levels,
                    "kernel_ms": round(ms, 4),
                    "kernel_timing": "HIP events on the launch stream between the launches, in a second pass of the same "
                                     "steps right after the timed region",
                    "kernels_ms": {k: round(v, 4) for k, v in ktimes},
                    "kernels_hbm_bytes_per_launch_live": ({k: round(v) for k, v in live.items()} if live else None),
                    "rhs": {"algorithmic_bytes_per_cell_level": algorithmic_bytes_per_cell_level(NT),
                            "ms": round(rhs_ms, 4), "kernels_sum_ms": round(kernels_sum_ms, 4), "achieved": round(rhs_ach, 1),
                            "frac": round(rhs_ach / HBM_PEAK_GBS, 4)}}

    # (everything the record reads exists from here on, whatever the stepping part reaches)
    wire = None
    wire_errors = None
    wire_check = None
    state_checksums = None
    nrk = args.rk4_steps if args.rk4_steps >= 0 else max(2, args.steps // 4)
    overlap = N > 1 and not (args.no_overlap or args.no_fuse_stages or args.unfused)
    emitted = []

    def mesh_int(name):     # (an older build of the library -- tools/ab_rounds.sh -- does not know every name)
        try:
            return mesh.get_int(name)
        except oa.OmegaAmdError:
            return None

    rhs_with_exchange = None
    realistic = None

    def emit(sypd, t_rk4, rk4_error, overlap_check, cpu=None):
        """rank 0: the ONE JSON line (also called by the watchdog below if the stepping part does not come back)"""
        if rank != 0 or emitted:
            return
        emitted.append(True)
        out = {"metric": "tendency_cell_level_updates_per_sec", "value": value, "unit": "cell-level-updates/s",
               "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
               "scaling_note": "value = RHS evaluations of the whole (N-way partitioned) mesh, every rank sweeping its owned + halo "
                               "cells, WITHOUT a halo exchange inside the timed loop (a tendency evaluation has none; "
                               "config.rhs_excludes_halo_exchange); the strong-scaling figure that includes the two exchanges of "
                               "a step is rk4.cell_level_updates_per_sec (= cells x levels x 4 evaluations / rk4.ms_per_step) "
                               "and sypd; rhs_with_halo_exchange = the same K evaluations, each preceded by the (not overlapped) exchange of its inputs",
               "data": "synthetic",
               "config": {"workload": desc, "cells": int(n_cells_global), "levels": K, "tracers": NT,
                          "boundary_edges": int(g["boundaryEdge"].sum()) if "boundaryEdge" in g else 0,
                          "kernel_paths": {f: mesh_int(f) for f in ("CellL1OK", "CellPVOK", "CellPVFinalOK", "Del2RingOK",
                                                                    "Del2VertOK", "NIrregularEdges", "MaxEdges", "DomM1",
                                                                    "NBadCells")},
                          "terms": "Default.yml (del2+del4, center fluxes)", "fused_rhs": not args.unfused,
                          "launcher": launcher,
                          "partition": f"{args.partition}{N}" + (f" (edge cut {edge_cut})" if N > 1 else ""), "halo_width": halo_width,
                          "halo_wire": "none (1 rank)" if N == 1 else
                          ("RCCL send/recv inside libomega_amd (" + json.dumps(comm.info()) + ")" if comm else
                           ((wire_note + (" " + json.dumps(wire.info()) if wire else "")) if wire_note else
                            "none: " + "; ".join(wire_errors or ["?"])[:300])),
                          "halo_wire_check": wire_check,
                          "rhs_excludes_halo_exchange": True,
                          "partition_independent": bool(N == 1 or halo_width >= 4), "mesh_order": "input " + ("hilbert" if args.block < 0 else "morton" if args.block == 0 else
                                                   "row-major" if args.block == 1 else f"blocked{args.block}")
                                        + ", local numbering by Decomp: " + args.local_order,
                          "device_ms_per_step": dev_ms / args.steps, "setup_s": round(setup_s, 1),
                          "untimed_settle": {"ms": args.settle_ms, "rhs_evaluations": settle_evals,
                                             "why": "load step -> ~25 ms power-management transient (profiles/r03_probe_step_ramp.json)"},
                          "hip_graph": graph_stats},
               "sypd": sypd, "rk4": {"steps": nrk, "dt_s": args.dt, "ms_per_step": None if t_rk4 is None else 1e3 * t_rk4,
                       "cell_level_updates_per_sec": None if t_rk4 is None else 4.0 * cell_levels / t_rk4,
                       "stage_updates": "separate kernels" if (args.no_fuse_stages or args.unfused) else "fused into the RHS kernels",
                       "halo_exchange": "none (1 rank)" if N == 1 else
                       ("overlapped with the stage's interior cells" if overlap else "after the stage"),
                       "error": rk4_error, "overlap_check": overlap_check,
                       "state_checksums_after_2_steps": state_checksums},
               "rhs_with_halo_exchange": rhs_with_exchange,
               "realistic": realistic,
               "roofline": roofline, "cpu_baseline": cpu}
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)
        os.dup2(2, 1)

    # N > 1: if the stepping part (wire set-up, RK4, cross-check) hangs -- a peer died, a collective never returns -- the
    # RHS measurement above must not be lost with it: rank 0 prints the record with rk4.error set and leaves.
    watchdog = None
    if N > 1 and rank == 0 and args.rk4_timeout > 0:
        import threading

        def give_up():
            emit(None, None, f"the stepping part did not come back within {args.rk4_timeout:.0f} s (halo wire set-up, RK4 steps "
                             "or the overlap cross-check hung); the RHS record stands", None)
            os._exit(4)
        watchdog = threading.Timer(args.rk4_timeout, give_up)
        watchdog.daemon = True
        watchdog.start()

    # ------------------------------------------------ the halo wire (N > 1): needed by the stepping part only
    wire = None
    wire_errors = None

    def gather_errors(err):
        errs = [None] * N
        dist.all_gather_object(errs, err)
        return [e for e in errs if e]

    def setup_rccl():
        """All ranks get a communicator, or none keeps one.  Returns the list of error strings (empty = success)."""
        nonlocal comm
        err = ""
        try:
            ident = [oa.RcclComm.unique_id() if rank == 0 else None]
        except Exception as exc:  # noqa: BLE001
            ident, err = [None], f"rank {rank}: ncclGetUniqueId: {exc}"
        dist.broadcast_object_list(ident, src=0)
        if ident[0] is not None:
            try:
                comm = oa.RcclComm(ident[0], N, rank)     # collective: ncclCommInitRank
            except Exception as exc:  # noqa: BLE001
                comm, err = None, f"rank {rank}: ncclCommInitRank: {exc}"
        errs = gather_errors(err)
        if errs and comm is not None:
            comm.abort()          # a communicator some ranks do not have must never carry traffic
            comm = None
        return errs

    def setup_peer():
        nonlocal wire
        err = ""
        try:
            rows = max(halo.recv_rows(1 + NT, 1, 0), 1)
            wire = oa.PeerWire(N, rank, rows * K * 8)
            handle = wire.handle()
        except Exception as exc:  # noqa: BLE001
            wire, handle, err = None, None, f"rank {rank}: PeerWire: {exc}"
        handles = [None] * N
        dist.all_gather_object(handles, handle)
        if err == "" and all(h is not None for h in handles):
            try:
                wire.connect(handles)
            except Exception as exc:  # noqa: BLE001
                err = f"rank {rank}: PeerWire.connect: {exc}"
        errs = gather_errors(err)
        if errs:
            wire = None
        return errs

    if N > 1:
        if args.backend in ("rccl", "auto"):
            wire_errors = setup_rccl()
            if not wire_errors:
                halo.use_rccl(comm)
            elif args.backend == "auto":
                rccl_errors = wire_errors
                wire_errors = setup_peer()
                if not wire_errors:
                    halo.use_peer(wire)
                    wire_note = ("peer copies (HIP IPC mailboxes + flag kernels inside libomega_amd, PeerWire.h); RCCL was "
                                 "tried first and could not be initialised: " + "; ".join(rccl_errors)[:300])
                else:
                    wire_errors = rccl_errors + wire_errors
        elif args.backend == "peer":
            wire_errors = setup_peer()
            if not wire_errors:
                halo.use_peer(wire)
                wire_note = "peer copies (HIP IPC mailboxes + flag kernels inside libomega_amd, PeerWire.h)"
        else:  # gloo rehearsal (opt-in above): host-staged messages, ranks may share one GPU
            from tests.gloo_transport import GlooStagedTransport
            GlooStagedTransport(halo)
            wire_note = "host-staged gloo (rehearsal, --allow-host-staged)"
        # The wire is checked before it carries the model state: the reference's HaloTest (test/base/HaloTest.cpp:41-100)
        # on the real wire -- owned cells and edges hold their global ids, one exchange, every halo layer must hold its
        # owners' ids.  A wire that fails is not replaced by another one: the cause is reported (rk4.error), the RHS
        # record stands, the run ends non-zero.
        wire_check = None
        if not wire_errors:
            err = ""
            try:
                for elem, ids, rows, n_own, n_all in ((0, cell_id, mesh.NCellsSize, mesh.NCellsOwned, mesh.NCellsAll),
                                                      (1, edge_id, mesh.NEdgesSize, mesh.NEdgesOwned, mesh.NEdgesAll)):
                    a = np.zeros(rows, dtype=np.int32)
                    a[:n_own] = ids[:n_own]
                    buf = oa.DeviceBuffer(a)
                    halo.exchange(buf.ptr, 1, rows, 1, elem, stream=stream, elem_bytes=4)
                    oa.device_synchronize()
                    if not np.array_equal(buf.to_host()[:n_all], np.asarray(ids[:n_all], dtype=np.int32)):
                        err = f"rank {rank}: wrong global ids in the halo after an exchange of element kind {elem}"
            except Exception as exc:  # noqa: BLE001
                err = f"rank {rank}: id exchange: {type(exc).__name__}: {exc}"
            errs = gather_errors(err)
            wire_check = "global-id exchange (cells, edges): " + ("ok on every rank" if not errs else "FAILED")
            if errs:
                wire_errors = ["the halo wire does not deliver the owners' values: " + "; ".join(errs)[:400]]
        if wire_errors and rank == 0:
            print("[bench] no halo wire: " + "; ".join(wire_errors), file=sys.stderr, flush=True)

    # ------------------------------------------------ RK4 steps (SYPD)
    nrk = args.rk4_steps if args.rk4_steps >= 0 else max(2, args.steps // 4)
    sypd = t_rk4 = rk4_error = None
    if wire_errors:     # strict: no wire, no stepping part; the RHS record stands and the run exits non-zero
        rk4_error = "no halo wire could be initialised: " + "; ".join(wire_errors)[:600]
        nrk = 0
    overlap = N > 1 and not (args.no_overlap or args.no_fuse_stages or args.unfused)
    if nrk > 0:
        if rank == 0:   # if a rank dies in the stepping part, the RHS measurement is at least in the log
            print(f"[bench] RHS: {ms_per_step:.4f} ms/step, {value:.4e} cell-level-updates/s on {N} GPU(s); "
                  f"RK4 ({'overlapped' if overlap else 'sequential'} exchanges) next", file=sys.stderr, flush=True)
        # No retry and no fall-back: a failure of the (overlapped) exchange is reported and the run ends non-zero.
        # RCCL work is never re-issued in a process whose exchange failed.
        try:
            stepper = oa.TimeStepper("RungeKutta4", args.dt, tend, aux, mesh, halo, tracers)
            stepper.set_option("FuseStageUpdates", not args.no_fuse_stages)
            stepper.set_option("OverlapHaloExchange", overlap)
            stepper.do_step(state, stream=stream)  # warm-up (allocations, RCCL connections)
            settle()                                # the wire set-up above left the GPU idle for seconds
            stepper.do_step(state, stream=stream)
            barrier()
            t1 = time.perf_counter()
            for _ in range(nrk):
                stepper.do_step(state, stream=stream)
            oa.device_synchronize()
            t_local = time.perf_counter() - t1
            barrier()
            t_rk4 = allmax(t_local) / nrk
            sypd = (args.dt / t_rk4) / 365.0
            hh, _ = state.copy_to_host(0)
            if not np.isfinite(hh[: mesh.NCellsOwned]).all():
                raise FloatingPointError("state went non-finite during the RK4 steps")
            if halo is not None:
                halo.check()    # the host is synchronised with the last exchange: a peer-wire wait that gave up shows now
        except Exception as exc:  # noqa: BLE001
            rk4_error = f"rank {rank}: {type(exc).__name__}: {exc}"
            sypd = t_rk4 = None
            print(f"[bench] RK4 FAILED on {rk4_error}", file=sys.stderr, flush=True)
        if N > 1:
            # the verdict is collective: every rank learns whether ANY rank failed, so that either all of them enter
            # the cross-check below or none does, and rank 0 prints the record before anybody leaves.  (A rank stuck
            # in a GPU wait because its peer died never gets here: the side channel's time limit ends the job.)
            all_errs = gather_errors(rk4_error)
            if all_errs and rk4_error is None:
                rk4_error = "; ".join(all_errs)[:600]
                sypd = t_rk4 = None

    # ------------------------------------------------ N > 1: the overlapped exchange against the sequential one
    # The one-GPU development boxes can only run the exchange logic over a host-staged wire, which synchronises the
    # streams; here, on the real wire, two short runs from the same initial state -- exchanges overlapped with the
    # interior work, and exchanges after the stage -- must leave the same bits (global double-double checksum of h, u
    # and the tracers over owned elements, combined in rank order).
    overlap_check = None

    def checksum(mode_overlap):
        """global double-double sums of h, u and every tracer over owned elements after two RK4 steps from the initial state"""
        ones_c, ones_e = oa.DeviceBuffer(np.ones(mesh.NCellsSize)), oa.DeviceBuffer(np.ones(mesh.NEdgesSize))
        state.copy_to_device(h, u, 0)
        upload_tracers(tracers)
        st2 = oa.TimeStepper("RungeKutta4", args.dt, tend, aux, mesh, halo, tracers)
        st2.set_option("OverlapHaloExchange", mode_overlap)
        for _ in range(2):
            st2.do_step(state, stream=stream)
        oa.device_synchronize()
        kp = oa.level_pitch(K)
        parts = [oa.local_weighted_sum_dd(ones_c.ptr, state.device_ptr(0, 0), mesh.NCellsOwned, K, row_pitch=kp, stream=stream),
                 oa.local_weighted_sum_dd(ones_e.ptr, state.device_ptr(1, 0), mesh.NEdgesOwned, K, row_pitch=kp, stream=stream)]
        parts += [oa.local_weighted_sum_dd(ones_c.ptr, tracers.device_ptr(0) + 8 * l * mesh.NCellsSize * kp,
                                           mesh.NCellsOwned, K, row_pitch=kp, stream=stream) for l in range(NT)]
        if N == 1:
            return [oa.combine_dd([p])[0] for p in parts]
        if comm is not None or wire is not None:    # globalSum inside the library, over the halo's own wire
            return [x for i in range(0, len(parts), 32) for x in halo.global_sum_dd(parts[i: i + 32], stream=stream)]
        return [oa.global_sum_dd(p) for p in parts]          # (host-staged rehearsal: over the side channel)

    # N = 1: the same sums, so that the records of an N = 1, 2, 4, 8 series can be compared with each other -- at
    # HaloWidth >= 4 the partitioned runs must give the one-rank run's sums (rk4.state_checksums_after_2_steps)
    state_checksums = None
    if N == 1 and nrk > 0 and rk4_error is None and not (args.no_fuse_stages or args.unfused):
        try:
            state_checksums = checksum(False)
        except Exception as exc:  # noqa: BLE001
            state_checksums = f"{type(exc).__name__}: {exc}"
    if N > 1 and nrk > 0 and rk4_error is None and overlap:
        check_err = None
        try:
            a, b = checksum(True), checksum(False)
            state_checksums = a
            overlap_check = {"overlapped_equals_sequential": a == b, "checksums_h_u_tracers": a}
            if a != b:
                overlap_check["sequential_checksums_h_u_tracers"] = b
            if a != b:   # the SYPD above was measured on an exchange that does not reproduce the sequential one
                rk4_error = "overlapped and sequential halo exchanges give different states (see rk4.overlap_check)"
                sypd = None
        except Exception as exc:  # noqa: BLE001
            check_err = f"rank {rank}: {type(exc).__name__}: {exc}"
        check_errs = gather_errors(check_err)
        if check_errs:
            overlap_check = {"error": "; ".join(check_errs)[:600]}
            rk4_error = "the overlapped-vs-sequential cross-check failed (see rk4.overlap_check)"
            sypd = None

    # ------------------------------------------------ N > 1: the evaluation WITH the exchange that feeds it
    # `value` above has no exchange in its timed loop (a tendency evaluation has none).  What a model pays per evaluation is
    # the halo update of its inputs first (RungeKutta4Stepper.cpp:95-101: h, u and the tracers after every stage): the same
    # K steps again, each one = exchange of h, u and the NT tracers on the launch stream, then the evaluation (the
    # sequential form: nothing overlapped, the upper bound of what the exchange costs).  Reported next to `value`.
    if N > 1 and nrk > 0 and rk4_error is None:
        x_err = None
        try:
            def one():
                halo.exchange_state(state, tracers, 0, stream=stream)
                tend.compute_all_tendencies(state, aux, tracers, stream=stream)
            for _ in range(args.warmup):
                one()
            barrier()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                one()
            oa.device_synchronize()
            t_x = time.perf_counter() - t1
            barrier()
            t_x = allmax(t_x) / args.steps
            halo.check()
            rhs_with_exchange = {"ms_per_step": 1e3 * t_x, "value": cell_levels / t_x, "unit": "cell-level-updates/s",
                                 "step": "the steppers' halo exchange (h, u and the tracers, one message per neighbour; on the launch "
                                         "stream, nothing overlapped), then computeAllTendencies; same W and K, barrier-bracketed, "
                                         "max over ranks"}
        except Exception as exc:  # noqa: BLE001
            x_err = f"rank {rank}: {type(exc).__name__}: {exc}"
        x_errs = gather_errors(x_err)
        if x_errs:
            rhs_with_exchange = {"error": "; ".join(x_errs)[:600]}

    if watchdog is not None:
        watchdog.cancel()

    # ------------------------------------------------ CPU baseline (rank 0, N = 1 only): the oracle
    cpu = None
    if N == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(nx, ny, K, NT, dc, args.dt) if (nx > 0 and not args.workload.endswith("_coast")) else None

    if N == 1 and rank == 0:
        wl = ("fib7_coast" if args.workload == "qu30" else "none") if args.realistic == "auto" else args.realistic
        if wl != "none":
            # (the headline problem's arrays are released first: configs[3] holds 5 GB, the block needs 2)
            try:
                del stepper
            except NameError:
                pass
            state = tracers = aux = tend = None
            import gc
            gc.collect()
            try:
                realistic = realistic_block(wl, args)
            except Exception as exc:  # noqa: BLE001  (the headline record stands)
                realistic = {"workload": wl, "error": f"{type(exc).__name__}: {exc}"}
    emit(sypd, t_rk4, rk4_error, overlap_check, cpu)
    if N > 1:
        # Ordered teardown, with a bound: the record is out; a rank must not hang in the destructors of RCCL / gloo / HIP
        # racing its peers' exits (that would keep the launcher -- and whoever timed it -- waiting).  Explicit closes while
        # every peer is still there, then the process leaves without interpreter finalisation; if a close itself does not
        # come back within a minute, the timer ends the process with the same code.
        import threading
        code = 3 if rk4_error else 0   # (3: the RHS record above stands; the stepping part failed and says so in rk4.error)
        bound = threading.Timer(60.0, lambda: os._exit(code))
        bound.daemon = True
        bound.start()
        oa.device_synchronize()
        dist.barrier()      # every rank's GPU work is complete: the wires may be taken down
        if wire is not None:
            wire.close()
        if comm is not None:
            comm.close()
        dist.barrier()
        dist.destroy_process_group()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(code)


def live_traffic(args):
    """roofline.traffic measured IN THIS RUN (N = 1): before this process loads the HIP runtime, two short child runs of
    this same script under `rocprofv3 --pmc` -- FETCH_SIZE in one, WRITE_SIZE in the other: the two do not fit one pass,
    and counters never share a run with the trace domains (MI355X_MICROARCH.md, HBM / rocprofv3 section) -- on the same
    workload, numbering and options.  HBM bytes per launch = FETCH_SIZE [KB] x 1024 x 2 (gfx950 tallies the 128-byte
    requests of 16-byte-per-lane loads at 64 B) + WRITE_SIZE [KB] x 1024, averaged over the launches of each kernel.
    The children are ordinary child processes started with the program itself after `--` (no exec from a process that
    has touched the GPU: this one has not yet).  Returns ({kernel: bytes per launch}, note) or (None, reason)."""
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None, "this process already runs under a profiler (its environment would reach the child runs)"
    from tools.summarise_profile import counters
    child = [sys.executable, os.path.abspath(__file__), "--workload", args.workload, "--steps", "3", "--warmup", "1", "--rk4-steps", "0",
             "--no-cpu-baseline", "--no-live-traffic", "--realistic", "none", "--settle-ms", "0", "--local-order", args.local_order,
             "--block", str(args.block), "--max-edges", str(args.max_edges), "--halo-width", str(args.halo_width)] + (["--unfused"] if args.unfused else [])
    # (forwarded: every argument that changes the mesh, its numbering or the kernels -- workload, local order, input
    # order, table width, halo width, fused / unfused; not forwarded: step counts, dt and the stepping part, which the
    # child runs do not have)
    work = tempfile.mkdtemp(prefix="omega_bench_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    got = {}
    t0 = time.time()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(work, ctr)
            with open(os.path.join(work, ctr + ".log"), "wb") as log:
                r = subprocess.run([exe, "--pmc", ctr, "--output-format", "csv", "-d", d, "-o", "p", "--", *child], env=env, cwd=ROOT,
                                   stdout=log, stderr=subprocess.STDOUT, timeout=120)
            if r.returncode != 0:
                return None, f"rocprofv3 --pmc {ctr} child run ended with code {r.returncode}"
            rows = counters(d)
            for k, v in rows.items():
                if v.get(ctr):
                    got.setdefault(k, {})[ctr] = (sum(v[ctr]) / len(v[ctr]), len(v[ctr]))
    except subprocess.TimeoutExpired:
        return None, "a rocprofv3 --pmc child run did not finish within 120 s (a pass takes ~ 10 s)"
    except Exception as exc:  # noqa: BLE001  (the measurement proper must not depend on the profiler)
        return None, f"{type(exc).__name__}: {exc}"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    out = {k: 2.0 * 1024.0 * v["FETCH_SIZE"][0] + 1024.0 * v["WRITE_SIZE"][0] for k, v in got.items()
           if "FETCH_SIZE" in v and "WRITE_SIZE" in v and "Body" in k}
    if not out:
        return None, "the PMC passes recorded none of the RHS kernels"
    n = min(v["FETCH_SIZE"][1] for k, v in got.items() if k in out)
    return out, (f"measured in this run: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in two child runs of `bench.py --workload "
                 f"{args.workload} --steps 3 --warmup 1` before the timed process touched the GPU ({time.time() - t0:.0f} s, >= {n} launches "
                 "per kernel); FETCH_SIZE x 2 (gfx950 half-count of 16-B-per-lane reads) + WRITE_SIZE")


def staged_bytes_exact(nc, ne, nv, nt, k):
    """SURVEY.md 8(d) B_staged in exact mesh counts: 8 K (10 NC + 5 NE + 7 NV + 5 NT NC) bytes per evaluation"""
    return 8 * k * (10 * nc + 5 * ne + 7 * nv + 5 * nt * nc)


def realistic_block(workload, args):
    """N = 1, untimed set-up + a short timed region of its own, AFTER the headline measurement: the fused RHS on a mesh
    shaped like the BASELINE configs' real ones -- a culled sphere with three valences (`fib7_coast`: relaxed Fibonacci
    lattice, pentagons / hexagons / heptagons, 28 % land removed, local numbering k-d) -- so that the driver-timed record
    carries a number that is not the friendliest mesh's.  Same W warm-up and K timed evaluations, HIP events on the
    launch stream; fractions on B_staged in EXACT element counts (a culled mesh has NE > 3 NC) and, for comparison with
    earlier rounds' tables, on the hexagon-count formula 8 (39 + 5 NT) per cell-level."""
    t0 = time.time()
    nx, ny, dc, K, NT, desc = WORKLOADS[workload]
    g = workload_mesh(workload)
    gm = oa.GlobalMesh(g)
    decomp = oa.Decomp(gm, 1, 0, 3, cell_task=None, local_order=args.local_order)
    mesh = oa.HorzMesh(decomp, K)
    cells0 = decomp.get_array("CellID")[: mesh.NCellsAll] - 1
    edges0 = decomp.get_array("EdgeID")[: mesh.NEdgesAll] - 1
    hh, uu, _ = synthetic_state_rows(g, K, 0, cells0, edges0, tracers=[])
    if "boundaryEdge" in g:
        uu[np.asarray(g["boundaryEdge"])[edges0] != 0] = 0.0
    h, u = np.zeros((mesh.NCellsSize, K)), np.zeros((mesh.NEdgesSize, K))
    h[: mesh.NCellsAll], u[: mesh.NEdgesAll] = hh, uu
    state = oa.OceanState(mesh, None, K, 2)
    tracers = oa.Tracers(mesh, None, K, NT, 2)
    aux = oa.AuxiliaryState(mesh, None, K, NT)
    tend = oa.Tendencies(mesh, K, NT, oa.default_config())
    state.copy_to_device(h, u, 0)
    kp = oa.level_pitch(K)
    for l in range(NT):
        t2 = synthetic_state_rows(g, K, NT, cells0, edges0[:0], tracers=[l])[2][0]
        buf = np.zeros((mesh.NCellsSize, kp))
        buf[: t2.shape[0], :K] = t2
        oa.copy_to_device(tracers.device_ptr(0) + 8 * l * mesh.NCellsSize * kp, buf)
    stream = oa.Stream()
    setup_s = time.time() - t0
    t1 = time.perf_counter()
    while args.settle_ms > 0 and (time.perf_counter() - t1) * 1e3 < args.settle_ms:
        for _ in range(4):
            tend.compute_all_tendencies(state, aux, tracers, stream=stream)
        oa.device_synchronize()
    for _ in range(args.warmup):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    oa.device_synchronize()
    ev0, ev1 = oa.Event(), oa.Event()
    tw = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    ev1.record(stream)
    oa.device_synchronize()
    wall_ms = 1e3 * (time.perf_counter() - tw) / args.steps
    dev_ms = ev0.elapsed_ms(ev1) / args.steps
    tend.kernel_timing(True)
    for _ in range(args.steps):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    oa.device_synchronize()
    tend.kernel_timing(False)
    ktimes = tend.collect_kernel_times()
    hT = tend.get(0)[: mesh.NCellsOwned]
    finite = bool(np.isfinite(hT).all() and np.abs(hT).max() > 0)
    b_exact = staged_bytes_exact(mesh.NCellsAll, mesh.NEdgesAll, mesh.NVerticesAll, NT, K)
    b_hex = algorithmic_bytes_per_cell_level(NT) * mesh.NCellsAll * K
    paths = {}
    for f in ("CellL1OK", "CellPVOK", "CellPVFinalOK", "Del2RingOK", "Del2VertOK", "NIrregularEdges", "MaxEdges", "DomM1",
              "NBadCells", "NWideCells", "NPatchTiles16", "NPatchFallback16"):
        try:
            paths[f] = mesh.get_int(f)
        except oa.OmegaAmdError:
            paths[f] = None
    return {"workload": workload + ": " + desc, "cells": int(g["nCells"]), "edges": int(g["nEdges"]), "vertices": int(g["nVertices"]),
            "levels": K, "tracers": NT, "boundary_edges": int(g["boundaryEdge"].sum()) if "boundaryEdge" in g else 0,
            "valences": {str(v): int(c) for v, c in enumerate(np.bincount(g["nEdgesOnCell"])) if c},
            "local_order": args.local_order, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": wall_ms, "device_ms_per_step": dev_ms,
            "value": g["nCells"] * K / (wall_ms * 1e-3), "unit": "cell-level-updates/s",
            "rhs": {"b_staged_bytes_exact_counts": b_exact, "achieved": round(b_exact / (dev_ms * 1e-3) / 1e9, 1),
                    "frac": round(b_exact / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                    "frac_hexagon_count_formula": round(b_hex / (dev_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "unit": "GB/s",
                    "peak": HBM_PEAK_GBS},
            "kernels_ms": {k: round(v, 4) for k, v in ktimes}, "kernel_paths": paths, "result_finite_nonzero": finite,
            "setup_s": round(setup_s, 1)}


def cpu_baseline(nx, ny, K, NT, dc, dt):
    """The CPU oracle (kind "port": restatement of the reference functors with the reference's launch structure,
    OpenMP over elements, built -O3 -march=native on this host) timed on the SAME mesh and state as the GPU run:
    a few RHS evaluations and one RK4 step (bounded to about 20-30 s of CPU work)."""
    from oracle import oracle as O       # (native build selected at the top of main)
    budget_cells = 500_000 * 80 * (8 + 2 * 6)      # about the QU30 workload: larger ones are sampled
    scale = 1
    while (nx // scale) * (ny // scale) * K * (8 + 2 * NT) > budget_cells:
        scale *= 2
    nxs, nys = nx // scale, ny // scale
    gs = planar_hex(nxs, nys, dc)
    if scale == 1:
        gs = reorder_cells_morton(gs)
    M = O.Mesh.single_rank(gs, K)
    hs, us, trs = synthetic_state(gs, K, NT)

    def pad(a):
        out = np.zeros(a.shape[:-2] + (a.shape[-2] + 1, a.shape[-1]))
        out[..., :-1, :] = a
        return out
    hs, us, trs = pad(hs), pad(us), pad(trs)
    cores = O.default_threads()
    O.lib().orc_set_num_threads(cores)
    orc = O.Oracle(M, NT)
    for _ in range(2):   # warm-ups (first touch of every aux array): SURVEY 8(d) asks for 2
        orc.compute_all_tendencies(hs, us, trs)
    # median of >= 10 RHS evaluations, bounded to about 16 s of CPU work
    times = []
    t0 = time.perf_counter()
    while len(times) < 10 or (len(times) < 20 and time.perf_counter() - t0 < 10.0):
        ta = time.perf_counter()
        orc.compute_all_tendencies(hs, us, trs)
        times.append(time.perf_counter() - ta)
        if time.perf_counter() - t0 > 16.0 and len(times) >= 5:
            break
    n = len(times)
    v = gs["nCells"] * K / float(np.median(times))
    # RK4: >= 3 steps when they fit into about 20 s, at least 1
    st = orc.make_state(hs, us, trs)
    steps = []
    t1 = time.perf_counter()
    while len(steps) < 3 and (not steps or time.perf_counter() - t1 + steps[-1] < 22.0):
        ta = time.perf_counter()
        orc.step("rk4", st, dt)
        steps.append(time.perf_counter() - ta)
    el2 = float(np.median(steps))
    ratio = (nx * ny) / gs["nCells"]
    sypd_cpu = (dt / (el2 * ratio)) / 365.0
    frac = "the full workload mesh" if scale == 1 else f"a 1/{scale * scale}-size mesh of the same shape"
    hc = O.host_cores()
    return {"value": v, "unit": "cell-level-updates/s", "cores": cores, "threads_used": cores,
            "cores_available": hc["cores_available"], "host": hc,
            "kind": "port" if scale == 1 else f"port, 1/{scale * scale} sample", "sypd": sypd_cpu, "rk4_steps": len(steps),
            "rhs_evaluations": n, "rhs_ms_median": 1e3 * float(np.median(times)), "rk4_ms_median": 1e3 * el2,
            "build": "gcc -O3 -march=native -ffp-contract=off -fopenmp (oracle/Makefile: native)",
            "sample": f"median of {n} RHS evaluations after 2 warm-ups and of {len(steps)} RK4 step(s) on {frac} ({nxs}x{nys} = "
                      f"{gs['nCells']} cells x {K}L x {NT} tracers), reference launch structure (23 passes), "
                      f"{cores} OpenMP threads (cap OMEGA_ORACLE_THREADS, default 16 = the one-GPU box's CPU share; the "
                      f"process may use {hc['cores_available']} of the host's {hc['logical_cpus']} logical CPUs)"
                      + ("" if scale == 1 else f"; sypd scaled by the cell ratio {ratio:.0f}")}


if __name__ == "__main__":
    main()
