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
bounded sample of the same workload on the host cores).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import omega_amd as oa  # noqa: E402
from omega_amd.meshgen import (icosahedral_points, planar_hex, reorder_cells_blocked, reorder_cells_morton,  # noqa: E402
                               spherical_voronoi, synthetic_state)

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
    "ico6": (0, 6, 0.0, 60, 2, "spherical icosahedral Voronoi mesh, 40962 cells (12 pentagons), 60L, 2 tracers"),
    "small": (96, 96, 30.0e3, 80, 6, "small smoke workload 96x96, 80L, 6 tracers"),
}


def algorithmic_bytes_per_cell_level(nt, kernel=None):
    """SURVEY.md 8(d) B_staged = 8*(39 + 5*NT) B per cell-level for the whole RHS (NE = 3NC,
    NV = 2NC); per kernel: the arrays that kernel must read / write once."""
    per_kernel = {
        "VortVertexBody": 8 * (1 + 3 + 2 * 2),                 # h, u -> RelVort, 1/LayerThickVertex
        "FusedCell1Body": 8 * (3 + 1 + nt + 3 + nt),            # u, h, tr -> KE, Div, hTend, Del2Tr
        "Del2CellRingBody": 8 * (1 + 2 + 1),                    # Div, RelVort -> Del2Div
        "Del2VertexSelBody": 8 * (1 + 2 + 2),                   # Div, RelVort -> Del2RelVort
        "CellPVBody<side 0>": 8 * (3 + 1 + 2 * 2 + 3),          # u, h, RelVort, InvThick -> running PV sums
        "CellPVFinalBody": 8 * (3 + 1 + 2 * 2 + 3 + 3 + 2 + 3),  # + sums, KE, Div, Del2Div, Del2RV -> uTend
        "FusedCell3Body": 8 * (nt + nt + 1 + 3 + nt),           # tr, Del2Tr, h, u -> trTend
    }
    per_kernel["CellPVBody<side 0>+FusedCell3Body"] = 8 * (nt + nt + 1 + 3 + nt + 2 * 2 + 3)
    per_kernel["FusedDel2CellBody"] = per_kernel["Del2CellRingBody"]
    per_kernel["FusedDel2VertexBody"] = per_kernel["Del2VertexSelBody"]
    per_kernel["CellPVBody<side 1>+EdgeFinalBody"] = per_kernel["CellPVFinalBody"] + 8 * (3 + 3 + 1)
    for k in ("FusedEdgeChainBody", "FusedEdgeBody", "edgePatchKernel"):
        per_kernel[k] = 8 * (3 + 1 + 3 * 2 + 3 + 2 + 3)         # u,h,3V,KE,Div,Del2Div,Del2RV -> uTend
    if kernel is None:
        return 8 * (39 + 5 * nt)
    return per_kernel[kernel]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("OMEGA_BENCH_WORKLOAD", "qu30"), choices=sorted(WORKLOADS))
    ap.add_argument("--rk4-steps", type=int, default=-1, help="RK4 steps for SYPD (default: max(2, steps//4))")
    ap.add_argument("--dt", type=float, default=600.0, help="time step [s] (Default.yml TimeStep 10 min)")
    ap.add_argument("--block", type=int, default=0, help="cell ordering of the synthetic mesh: 0 = Morton curve (default), -1 = Hilbert curve, 1 = row-major, n > 1 = n x n blocks")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--unfused", action="store_true", help="time the reference-structured launch sequence instead")
    ap.add_argument("--no-overlap", action="store_true",
                    help="N > 1: exchange halos after the producing stage instead of overlapped with its interior part")
    ap.add_argument("--no-fuse-stages", action="store_true",
                    help="RK4 with the separate update kernels instead of stage updates folded into the RHS kernels")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL over xGMI (production); gloo = host-staged rehearsal of the N>1 code path")
    ap.add_argument("--single-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    N = world

    nx, ny, dc, K, NT, desc = WORKLOADS[args.workload]
    dist = torch = None
    stream = None
    if N > 1:
        import torch
        import torch.distributed as dist
        if args.single_device:
            local_rank = 0
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    oa.device_init(local_rank)

    t0 = time.time()
    if args.workload.startswith("ico"):   # sphere: cells already numbered along a Morton curve in (lon, z)
        g = spherical_voronoi(points=icosahedral_points(ny), lloyd=0)
    else:
        g = planar_hex(nx, ny, dc)
    if args.workload.startswith("ico"):
        pass
    elif args.block <= 0:
        g = reorder_cells_morton(g, hilbert=args.block < 0)
    elif args.block > 1:
        g = reorder_cells_blocked(g, args.block)
    gm = oa.GlobalMesh(g)
    decomp = oa.Decomp(gm, N, rank, 3)
    mesh = oa.HorzMesh(decomp, K)
    halo = oa.Halo(decomp) if N > 1 else None
    cell_id, edge_id = decomp.get_array("CellID"), decomp.get_array("EdgeID")
    hg, ug, trg = synthetic_state(g, K, NT)

    def to_local(glob, ids, rows):
        out = np.zeros(glob.shape[:-2] + (rows, glob.shape[-1]))
        out[..., : rows - 1, :] = glob[..., ids[: rows - 1] - 1, :]
        return out

    h = to_local(hg, cell_id, mesh.NCellsSize)
    u = to_local(ug, edge_id, mesh.NEdgesSize)
    tr = to_local(trg, cell_id, mesh.NCellsSize)
    del hg, ug, trg

    if N > 1 and args.backend == "nccl":
        tstream = torch.cuda.Stream()
        stream = oa.Stream(handle=tstream.cuda_stream)
        from omega_amd.transport import TorchTransport
        transport = TorchTransport(halo, per_cell=K * (1 + NT), per_edge=K, device=f"cuda:{local_rank}", stream=tstream)
    elif N > 1:  # gloo rehearsal: host-staged messages, default stream (torch's .cpu() orders on it)
        stream = None
        from omega_amd.transport import TorchTransport
        transport = TorchTransport(halo, per_cell=K * (1 + NT), per_edge=K, device=f"cuda:{local_rank}", stream=None)
    else:
        stream = oa.Stream()

    cfg = oa.default_config()
    state = oa.OceanState(mesh, halo, K, 2)
    tracers = oa.Tracers(mesh, halo, K, NT, 2)
    aux = oa.AuxiliaryState(mesh, halo, K, NT)
    tend = oa.Tendencies(mesh, K, NT, cfg)
    tend.set_fused(not args.unfused)
    state.copy_to_device(h, u, 0)
    tracers.copy_to_device(tr, 0)
    setup_s = time.time() - t0

    def barrier():
        oa.device_synchronize()
        if N > 1:
            dist.barrier()
            oa.device_synchronize()

    def allmax(x):
        if N == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=f"cuda:{local_rank}" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    # ------------------------------------------------ RHS: W warm-up + K timed steps
    for _ in range(args.warmup):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    barrier()
    tend.kernel_timing(True)
    ev0, ev1 = oa.Event(), oa.Event()
    t_start = time.perf_counter()
    ev0.record(stream)
    for _ in range(args.steps):
        tend.compute_all_tendencies(state, aux, tracers, stream=stream)
    ev1.record(stream)
    barrier()
    wall_rhs = time.perf_counter() - t_start
    dev_ms = ev0.elapsed_ms(ev1)
    tend.kernel_timing(False)
    ktimes = tend.collect_kernel_times()
    wall_rhs = allmax(wall_rhs)
    ms_per_step = 1e3 * wall_rhs / args.steps

    n_cells_global = g["nCells"]
    cell_levels = n_cells_global * K
    value = cell_levels / (wall_rhs / args.steps)

    # ------------------------------------------------ RK4 steps (SYPD)
    nrk = args.rk4_steps if args.rk4_steps >= 0 else max(2, args.steps // 4)
    sypd = t_rk4 = rk4_error = None
    overlap = N > 1 and not (args.no_overlap or args.no_fuse_stages or args.unfused)
    if nrk > 0:
        # The RHS number above must survive a problem in the stepping part (the multi-GPU exchange path
        # cannot be rehearsed on the one-GPU development boxes): a deterministic failure of the overlapped
        # exchange falls back to the sequential one, a failure of that is reported in the JSON line.
        for attempt in (0, 1):
            try:
                stepper = oa.TimeStepper("RungeKutta4", args.dt, tend, aux, mesh, halo, tracers)
                stepper.set_option("FuseStageUpdates", not args.no_fuse_stages)
                stepper.set_option("OverlapHaloExchange", overlap)
                stepper.do_step(state, stream=stream)  # warm-up (allocations, RCCL connections)
                barrier()
                t1 = time.perf_counter()
                for _ in range(nrk):
                    stepper.do_step(state, stream=stream)
                barrier()
                t_rk4 = allmax(time.perf_counter() - t1) / nrk
                sypd = (args.dt / t_rk4) / 365.0
                hh, _ = state.copy_to_host(0)
                assert np.isfinite(hh[: mesh.NCellsOwned]).all(), "state went non-finite during the RK4 steps"
                rk4_error = None
                break
            except Exception as exc:  # noqa: BLE001
                rk4_error = f"{type(exc).__name__}: {exc}"
                sypd = t_rk4 = None
                if not overlap:
                    break
                overlap = False

    # ------------------------------------------------ roofline of the dominant kernel (rank 0's view)
    roofline = None
    if ktimes:
        name, ms = max(ktimes, key=lambda kv: kv[1])
        local_cell_levels = mesh.NCellsAll * K  # every launch sweeps owned + halo elements
        ach = algorithmic_bytes_per_cell_level(NT, name) * local_cell_levels / (ms * 1e-3) / 1e9
        rhs_ms = sum(m for _, m in ktimes)
        rhs_ach = algorithmic_bytes_per_cell_level(NT) * local_cell_levels / (rhs_ms * 1e-3) / 1e9
        # HBM bytes per launch of that kernel from the committed PMC passes (rocprofv3 --pmc FETCH_SIZE /
        # WRITE_SIZE on this same command, gfx950 correction applied; profiles/): only valid for the
        # workload and partition they were collected on
        traffic = traffic_src = None
        pmc_file = os.path.join(ROOT, "profiles", "r01_v7_bench_qu30_pmc.json")
        if args.workload == "qu30" and N == 1 and not args.unfused and os.path.exists(pmc_file):
            with open(pmc_file) as fh:
                pmc = json.load(fh)
            base = name.split("<")[0].split("+")[0]
            cands = [k for k, rec in pmc.items() if isinstance(rec, dict) and k.split("<")[0] == base]
            # `..., true>` instantiations are the RK4 stage-fused variants; the RHS timed here is the plain one
            plain = [k for k in cands if not k.endswith(", true>")] or cands
            if plain:
                traffic, traffic_src = pmc[plain[0]]["hbm_bytes_per_launch"], "profiles/r01_v7_bench_qu30_pmc.json"
        roofline = {"bound": "hbm", "kernel": name, "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_unit": "bytes/launch",
                    "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": algorithmic_bytes_per_cell_level(NT, name) * local_cell_levels,
                    "kernel_ms": round(ms, 4),
                    "kernels_ms": {k: round(v, 4) for k, v in ktimes},
                    "rhs": {"algorithmic_bytes_per_cell_level": algorithmic_bytes_per_cell_level(NT),
                            "ms": round(rhs_ms, 4), "achieved": round(rhs_ach, 1),
                            "frac": round(rhs_ach / HBM_PEAK_GBS, 4)}}

    # ------------------------------------------------ CPU baseline (rank 0, N = 1 only): the oracle
    cpu = None
    if N == 1 and rank == 0 and not args.no_cpu_baseline:
        cpu = cpu_baseline(K, NT, dc)

    if rank == 0:
        out = {"metric": "tendency_cell_level_updates_per_sec", "value": value, "unit": "cell-level-updates/s",
               "n_gpus": N, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
               "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64",
               "data": "synthetic",
               "config": {"workload": desc, "cells": int(n_cells_global), "levels": K, "tracers": NT,
                          "terms": "Default.yml (del2+del4, center fluxes)", "fused_rhs": not args.unfused,
                          "partition": f"rcb{N}", "halo_width": 3, "mesh_order": "hilbert" if args.block < 0 else "morton" if args.block == 0 else f"blocked{args.block}",
                          "device_ms_per_step": dev_ms / args.steps, "setup_s": round(setup_s, 1)},
               "sypd": sypd, "rk4": {"steps": nrk, "dt_s": args.dt, "ms_per_step": None if t_rk4 is None else 1e3 * t_rk4,
                       "stage_updates": "separate kernels" if (args.no_fuse_stages or args.unfused) else "fused into the RHS kernels",
                       "halo_exchange": "none (1 rank)" if N == 1 else
                       ("overlapped with the stage's interior cells" if overlap else "after the stage"),
                       "error": rk4_error},
               "roofline": roofline, "cpu_baseline": cpu}
        print(json.dumps(out))
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline(K, NT, dc):
    """The CPU oracle (kind "port": restatement of the reference functors, OpenMP over elements)
    timed on a bounded sample: a 1/16-size mesh of the same shape, same levels / tracers."""
    from oracle import oracle as O
    nxs = nys = 170
    gs = planar_hex(nxs, nys, dc)
    M = O.Mesh.single_rank(gs, K)
    hs, us, trs = synthetic_state(gs, K, NT)

    def pad(a):
        out = np.zeros(a.shape[:-2] + (a.shape[-2] + 1, a.shape[-1]))
        out[..., :-1, :] = a
        return out
    hs, us, trs = pad(hs), pad(us), pad(trs)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))  # a one-GPU box's CPU share is 16 cores
    O.lib().orc_set_num_threads(cores)
    orc = O.Oracle(M, NT)
    orc.compute_all_tendencies(hs, us, trs)  # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        orc.compute_all_tendencies(hs, us, trs)
        n += 1
        el = time.perf_counter() - t0
        if el > 10.0 or n >= 50:
            break
    v = gs["nCells"] * K * n / el
    # SYPD of the CPU path: RK4 steps of the oracle on the same sample, scaled by the cell ratio
    st = orc.make_state(hs, us, trs)
    orc.step("rk4", st, 600.0)  # warm-up
    nst, t1 = 0, time.perf_counter()
    while True:
        orc.step("rk4", st, 600.0, sim_time=600.0 * (nst + 1))
        nst += 1
        el2 = time.perf_counter() - t1
        if el2 > 5.0 or nst >= 10:
            break
    ratio = WORKLOADS["qu30"][0] * WORKLOADS["qu30"][1] / gs["nCells"]
    sypd_cpu = (600.0 / (el2 / nst * ratio)) / 365.0
    return {"value": v, "unit": "cell-level-updates/s", "cores": cores, "kind": "port",
            "sypd": sypd_cpu, "rk4_steps": nst,
            "sample": f"{n} RHS evaluations and {nst} RK4 steps of a {nxs}x{nys}-cell ({gs['nCells']} cells = 1/16 of the "
                      f"workload) x {K}L x {NT} tracers mesh, reference launch structure (23 passes), OpenMP threads = "
                      f"cores; sypd = dt 600 s / (sample step time x 16)"}


if __name__ == "__main__":
    main()
