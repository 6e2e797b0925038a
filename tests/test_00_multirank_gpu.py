"""2-rank run of the full product path on the GPU: C++ Decomp / Halo with HIP pack / unpack
kernels, RungeKutta4Stepper::doStep with its two exchange points, one process per rank, both on
GPU 0 (this pool's test boxes have one GPU), messages staged through gloo (tests/gloo_transport.py
test mode).  Must reproduce the single-rank CPU oracle bit for bit on owned elements.

The ranks (and the re-runs of the parity suite under other kernel structures) are fresh child processes; nothing in
this file touches the GPU in the test runner's own process.  tests/conftest.py collects the child-spawning files
first, so that the runner has not initialised the GPU yet when they start (rule of the pool: no exec in a process
that has touched the GPU -- fresh children are fine, but keeping the runner GPU-free while they run also keeps the
number of processes on the card at ranks + 0).
"""
import os
import subprocess
import sys

import pytest

from tests.test_multirank_cpu import ROOT, run_ranks

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("extra", [
    ["--no-del4"],
    ["--halo-width", 4, "--nx", 24, "--ny", 24, "--levels", 6],
    ["--no-del4", "--stepper", "Forward-Backward", "--levels", 5, "--tracers", 1],
    ["--no-del4", "--nx", 64, "--ny", 16, "--levels", 4],                   # band AND interior launches, overlapped
    ["--no-del4", "--nx", 64, "--ny", 16, "--levels", 4, "--no-overlap"],   # same, exchange after the stage
    ["--halo-width", 4, "--nx", 72, "--ny", 12, "--levels", 3, "--steps", 3],
    ["--no-del4", "--nx", 64, "--ny", 16, "--levels", 4, "--tracers", 0],   # no tracer kernel: exchange starts after the u band
    ["--no-del4", "--mesh", "ico3", "--levels", 4],                          # sphere, pentagon ring launches, 2 ranks
    ["--halo-width", 4, "--nx", 48, "--ny", 24, "--levels", 6, "--local-order", "curve"],   # Morton-ordered local numbering
    ["--halo-width", 4, "--mesh", "ico4", "--levels", 6, "--partition", "graph", "--local-order", "curve"],  # sphere: pentagon
                                                                  # lists next to the paired launches, del4 on, overlapped
    ["--halo-width", 4, "--mesh", "fib1500", "--levels", 6, "--partition", "graph", "--local-order", "curve"],  # pentagons,
                                     # hexagons and heptagons, 7-wide tables: the sweeps take valence 6, band / interior lists
    ["--halo-width", 4, "--nx", 48, "--ny", 24, "--levels", 20],     # 20 levels: device rows padded to 32 (pack / unpack with a pitch)
    ["--halo-width", 4, "--nx", 64, "--ny", 24, "--levels", 4, "--user-stream"],               # non-blocking user stream,
    ["--halo-width", 4, "--nx", 64, "--ny", 24, "--levels", 4, "--user-stream", "--no-overlap"],  # overlapped and sequential
    # partition lines crossing a coast (culled meshes): band / interior lists, irregular-edge list, masked boundary edges
    ["--halo-width", 4, "--mesh", "hex48x24_coast_mixed", "--levels", 6, "--partition", "graph", "--local-order", "curve"],
    ["--halo-width", 4, "--mesh", "hex64x16_coast_strait_raw", "--levels", 4, "--no-overlap"],
    ["--halo-width", 4, "--mesh", "ico4_coast_lakes", "--levels", 6, "--partition", "graph", "--local-order", "curve"],
    ["--no-del4", "--mesh", "fib1500_coast_ragged", "--levels", 4, "--stepper", "Forward-Backward"],
])
def test_two_ranks_one_gpu(extra):
    outs = run_ranks("gpu", 2, extra, timeout=900)
    assert all("OK" in o for o in outs)


@pytest.mark.parametrize("world,extra", [
    (2, ["--halo-width", 4, "--nx", 64, "--ny", 24, "--levels", 4, "--tracers", 6]),                  # overlapped
    (2, ["--halo-width", 4, "--nx", 64, "--ny", 24, "--levels", 4, "--tracers", 6, "--no-overlap"]),   # sequential
    (2, ["--halo-width", 4, "--nx", 64, "--ny", 24, "--levels", 4, "--user-stream"]),       # non-blocking user stream
    (2, ["--no-del4", "--stepper", "Forward-Backward", "--levels", 5, "--tracers", 1]),
    (2, ["--halo-width", 4, "--nx", 48, "--ny", 24, "--levels", 20, "--steps", 4]),         # padded rows, more exchanges
    (2, ["--halo-width", 4, "--mesh", "ico4_coast_lakes", "--levels", 6, "--partition", "graph", "--local-order", "curve"]),
    (4, ["--halo-width", 4, "--nx", 48, "--ny", 48, "--levels", 3, "--tracers", 6]),        # several neighbours per rank
    (4, ["--halo-width", 4, "--nx", 48, "--ny", 48, "--levels", 3, "--tracers", 6, "--no-overlap"]),
    (5, ["--halo-width", 4, "--nx", 60, "--ny", 48, "--levels", 4, "--tracers", 3, "--partition", "graph",
         "--local-order", "curve"]),
    # EddyDiff4 != 0 (Default.yml has the term enabled with coefficient 0): the tracers' radius-2 term then reaches two
    # halo layers far, which the sweep lengths of the RK4 stages (StageUpdate::NCellsTr) have to cover
    (2, ["--halo-width", 4, "--nx", 64, "--ny", 24, "--levels", 4, "--tracers", 3, "--eddy-diff4", 1.0e11]),
    (4, ["--halo-width", 4, "--mesh", "ico4", "--levels", 3, "--tracers", 2, "--eddy-diff4", 1.0e11, "--partition", "graph",
         "--local-order", "curve"]),
    # (r4) k-d local numbering; cells out of ring order (MeshView::BadCells) next to the partition line and in the halo
    (3, ["--halo-width", 4, "--mesh", "ico4", "--levels", 6, "--partition", "graph", "--local-order", "kd"]),
    (2, ["--halo-width", 4, "--mesh", "hex48x24_perm5", "--levels", 6, "--tracers", 3, "--partition", "graph", "--local-order", "kd"]),
])
def test_peer_wire_stream_ordered_exchanges(world, extra):
    """The same runs over the library's OTHER wire (PeerWire: HIP IPC mailboxes, device-to-device copies and flag
    kernels on the exchange's stream -- no host synchronisation anywhere, unlike the host-staged gloo rig): the
    overlapped stages' event ordering (band final -> communication stream -> halo in place -> next consumer) is
    exercised for real, and must give the single-rank oracle's bits in overlapped and in sequential mode."""
    outs = run_ranks("gpu", world, [*extra, "--wire", "ipc"], timeout=900)
    assert all("OK" in o and "peer wire" in o for o in outs)


@pytest.mark.parametrize("world", [2, 4])
def test_peer_wire_under_back_to_back_exchanges(world):
    """150 exchanges of changing data queued back to back on two alternating non-blocking streams, every result
    verified: the flag protocol (consumed / arrived counters, release stores, acquire loads on uncached memory) and the
    reuse of the one mailbox per rank under pressure."""
    outs = run_ranks("gpu", world, ["--wire", "ipc", "--stress", 150, "--halo-width", 4, "--nx", 48, "--ny", 48,
                                    "--levels", 8, "--steps", 1], timeout=900)
    assert all("OK" in o and "peer wire" in o for o in outs)


def test_peer_wire_gives_up_on_a_silent_peer():
    """A rank whose neighbour never takes part in an exchange: the wait kernel leaves after the wire's time limit
    (every wave reaches its exit), the status is sticky and the next exchange fails with a message."""
    outs = run_ranks("gpu", 2, ["--wire", "ipc", "--peer-timeout-test"], timeout=300)
    assert all("OK (peer wire timeout)" in o for o in outs)


def test_halo_width_3_with_del4_bound_on_the_gpu():
    """bench.py's former N > 1 setting (reference default HaloWidth 3 + del4): bounded, non-zero deviation from
    the single-rank oracle (see tests/test_multirank_cpu.py)."""
    from tests.test_multirank_cpu import HALO3_DEL4_BOUND
    outs = run_ranks("gpu", 2, ["--nx", 24, "--ny", 24, "--rtol", HALO3_DEL4_BOUND], timeout=900)
    assert all("OK" in o and "max deviation" in o for o in outs)


HALO3 = ["--halo-width", 3, "--against-partitioned", "--eddy-diff4", 1.0e11]


@pytest.mark.parametrize("options", ["", "SendBand=0,BandOnComm=0,ShrinkSweeps=0"])
@pytest.mark.parametrize("world,extra", [
    (2, [*HALO3, "--nx", 24, "--ny", 24, "--tracers", 2]),                                  # overlapped, host-staged wire
    (2, [*HALO3, "--nx", 24, "--ny", 24, "--tracers", 2, "--no-overlap"]),                  # sequential
    (2, [*HALO3, "--nx", 64, "--ny", 24, "--levels", 4, "--tracers", 6, "--wire", "ipc"]),  # band AND interior launches
    (4, [*HALO3, "--nx", 48, "--ny", 48, "--levels", 3, "--tracers", 3, "--wire", "ipc"]),
    (4, [*HALO3, "--nx", 48, "--ny", 48, "--levels", 3, "--tracers", 3, "--wire", "ipc", "--no-overlap"]),
    (4, [*HALO3, "--mesh", "ico4", "--levels", 3, "--tracers", 2, "--partition", "graph", "--local-order", "kd", "--wire", "ipc"]),
    (2, [*HALO3, "--mesh", "hex48x24_coast_mixed", "--levels", 6, "--partition", "graph", "--local-order", "kd"]),
    # the headline's shape per rank: 80 levels (five level chunks per workgroup), 6 tracers (LDS tile patches), k-d numbering
    (2, [*HALO3, "--nx", 64, "--ny", 32, "--levels", 80, "--tracers", 6, "--local-order", "kd", "--wire", "ipc"]),
    (4, [*HALO3, "--nx", 96, "--ny", 96, "--levels", 16, "--tracers", 6, "--partition", "graph", "--local-order", "kd", "--wire", "ipc"]),
    # a halo narrower than the reference's default: still "whatever the reference prints" (every shortcut keyed on the halo
    # width must fall back to full sweeps)
    (2, ["--halo-width", 2, "--against-partitioned", "--eddy-diff4", 1.0e11, "--nx", 32, "--ny", 24, "--tracers", 2]),
    # three valences (pentagons, hexagons, heptagons: narrow tables + wide-cell lists) on three ranks
    (3, [*HALO3, "--mesh", "fib1500", "--levels", 4, "--tracers", 2, "--partition", "graph", "--local-order", "kd", "--wire", "ipc"]),
    # the other two schemes exchange once per step (RungeKutta2Stepper.cpp:27-73: two evaluations in between;
    # ForwardBackwardStepper.cpp:27-82: the velocity evaluation reads the updated thickness)
    (2, [*HALO3, "--nx", 24, "--ny", 24, "--tracers", 2, "--stepper", "RungeKutta2"]),
    (2, [*HALO3, "--nx", 24, "--ny", 24, "--tracers", 2, "--stepper", "Forward-Backward", "--wire", "ipc"]),
])
def test_reference_default_halo_width_3_reproduces_the_partitioned_reference_run(world, extra, options, monkeypatch):
    """The reference's own N > 1 configuration (Default.yml:15 HaloWidth 3, del4 on, RK4 exchanging after every second
    evaluation, RungeKutta4Stepper.cpp:107-113): its result depends on the partition -- the second evaluation after an
    exchange reads halo rows the first one computed from incomplete stencils.  What the product has to print there is
    what Omega prints at the same N: the PARTITIONED oracle (each rank's oracle on its local mesh, sweeping NCellsAll /
    NEdgesAll / NVerticesAll like Tendencies.cpp:281-564 and TimeStepper.cpp:395-520, exchanging at the reference's two
    points) -- every local element, owned and halo, bit for bit, in overlapped and sequential mode, with the stage
    shortcuts (SendBand / ShrinkSweeps) on and off."""
    if options:
        monkeypatch.setenv("OMEGA_AMD_OPTIONS", options)
    outs = run_ranks("gpu", world, extra, timeout=900)
    assert all("OK" in o and "equal to the partitioned oracle" in o for o in outs)
    # ... and those bits ARE partition dependent here (otherwise the case would show nothing the HaloWidth 4 cases do not)
    import re
    dev = [float(re.search(r"deviation from the 1-rank run ([0-9.eE+-]+)", o).group(1)) for o in outs]
    narrow = extra[extra.index("--halo-width") + 1] < 3      # (HaloWidth 2: measured 1.7e-5 after two steps)
    assert max(dev) < (1.0e-3 if narrow else 1.0e-5), dev
    if "--stepper" not in extra:        # (RK4: two radius-2 evaluations between exchanges need more than 3 layers)
        assert max(dev) > 0.0, dev


@pytest.mark.parametrize("world,extra", [
    (2, ["--halo-width", 4, "--nx", 64, "--ny", 24, "--levels", 4, "--tracers", 6]),                 # overlapped
    (2, ["--halo-width", 4, "--nx", 64, "--ny", 24, "--levels", 4, "--tracers", 6, "--no-overlap"]),
    (4, ["--halo-width", 4, "--nx", 48, "--ny", 48, "--levels", 3, "--tracers", 6]),
])
def test_bench_setting_halo_width_4_with_del4_is_bit_exact(world, extra):
    """What `bench.py --gpus N` runs: HaloWidth 4, every Default.yml term incl. del4, 6 tracers, RK4 with
    stage-fused updates and (un)overlapped exchanges -- owned elements equal the single-rank oracle bit for bit."""
    outs = run_ranks("gpu", world, extra, timeout=900)
    assert all("OK" in o for o in outs)


def test_generic_fallback_kernels_in_a_child_process():
    """Meshes whose EdgesOnEdge / EdgesOnCell lists are not in MPAS ring order take the generic kernels
    (FusedEdgeBody, FusedDel2CellBody, FusedDel2VertexBody, separate update sweeps).  No generated mesh
    is like that, so the option ForceGeneric = 1 (omg_set_option, applied by omega_amd/__init__.py from OMEGA_AMD_OPTIONS in
    the child) clears the ring-table flags and the parity tests run again."""
    env = dict(os.environ, OMEGA_AMD_OPTIONS="ForceGeneric=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-k",
                        "(compute_all_tendencies and fused and (K80 or K4_ or K5 or ico3 or coast)) or time_steppers or generic_flags"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert " passed" in out and "failed" not in out


def test_file_width_kernels_in_a_child_process():
    """Option KeepMaxEdges = 1: HorzMesh keeps the file's maxEdges instead of the largest valence present, so the
    icosahedral mesh stored with maxEdges = 8 runs the 8-wide kernel instantiations with its hexagons as a "rarer
    valence" and its pentagons' edges on the edge-centric list -- what a mesh with real 8-valent cells would do."""
    env = dict(os.environ, OMEGA_AMD_OPTIONS="KeepMaxEdges=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-k",
                        "ico3pad8 or sphere_meshes_take_the_fast_paths or rk4_on_the_sphere"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert " passed" in out and "failed" not in out


def test_five_ranks_graph_partition_curve_order_one_gpu():
    """Five ranks on one card (the pool allows six processes on a GPU; one is left for the test runner): graph partition, Morton-ordered local numbering, HaloWidth 4, every
    Default.yml term, overlapped exchanges -- several neighbours per rank in one pack / unpack launch each."""
    outs = run_ranks("gpu", 5, ["--halo-width", 4, "--nx", 60, "--ny", 48, "--levels", 4, "--tracers", 3,
                                "--partition", "graph", "--local-order", "curve"], timeout=900)
    assert all("OK" in o for o in outs)


def test_unmerged_unpaired_kernel_structure_in_a_child_process():
    """Options MergeL1 = 0, Pair = 0: the seven separate kernels of round 1 (what meshes without the cell-side vertex
    tables fall back to) must still equal the oracle."""
    env = dict(os.environ, OMEGA_AMD_OPTIONS="MergeL1=0,Pair=0")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-k",
                        "(compute_all_tendencies and fused and (K80 or K4_ or K60 or ico3 or fib1500 or coast)) or time_steppers or rk4_on_the_sphere"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert " passed" in out and "failed" not in out


def test_wide_tables_only_in_a_child_process():
    """Option NarrowTables = 0: meshes of hexagons with a few heptagons keep ONE set of cell tables, 7 wide, and sweep
    them with the 6-valent kernel instantiations (round 2's structure; what meshes whose ring tables are not all valid
    still do).  Default: a second, 6-wide set for the sweeps + list launches of the 7-slot kernels for the heptagons."""
    env = dict(os.environ, OMEGA_AMD_OPTIONS="NarrowTables=0")
    r = subprocess.run([sys.executable, "-m", "pytest", "tests/test_gpu_parity.py", "-x", "-q", "-k",
                        "fib1500 or fib300 or sphere_meshes_take_the_fast_paths"],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
    out = r.stdout.decode()
    assert r.returncode == 0, out[-3000:]
    assert " passed" in out and "failed" not in out


@pytest.mark.parametrize("options", ["SendBand=0,BandOnComm=0,ShrinkSweeps=0", "SendBand=1,BandOnComm=0,ShrinkSweeps=0",
                                     "SendBand=0,BandOnComm=1,ShrinkSweeps=1"])
def test_rk4_stages_without_the_halo_shortcuts(options, monkeypatch):
    """Options SendBand / BandOnComm / ShrinkSweeps off: every kernel of every RK4 stage sweeps all local cells and the
    band launches stay on the compute stream (round 2's structure) -- the same bits, on both wires."""
    monkeypatch.setenv("OMEGA_AMD_OPTIONS", options)
    base = ["--halo-width", 4, "--nx", 64, "--ny", 24, "--levels", 4, "--tracers", 3, "--eddy-diff4", 1.0e11]
    outs = run_ranks("gpu", 2, [*base, "--wire", "ipc"], timeout=900)
    assert all("OK" in o and "peer wire" in o for o in outs)
    outs = run_ranks("gpu", 2, base, timeout=900)
    assert all("OK" in o for o in outs)


def test_four_ranks_one_gpu():
    """Four ranks (2 x 2 blocks of a 48 x 48 mesh: every rank has several neighbours, corner halos travel
    through two of them) on one GPU, overlapped exchanges."""
    outs = run_ranks("gpu", 4, ["--no-del4", "--nx", 48, "--ny", 48, "--levels", 3], timeout=900)
    assert all("OK" in o for o in outs)


def test_bench_with_two_ranks_on_one_gpu():
    """bench.py's own N > 1 path, rehearsed the way tools/rehearse_n.sh does it: two ranks under torch.distributed.run on
    ONE GPU (RCCL refuses that, so `auto` takes the library's peer wire), the small workload.  The record must carry the
    wire check, a stepping part without error, overlapped == sequential, and the state sums of the one-rank run."""
    import json
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--workload", "small", "--no-cpu-baseline", "--no-live-traffic"]
    r1 = subprocess.run([*base, "--gpus", "1"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r1.returncode == 0, r1.stderr.decode()[-2000:]
    one = json.loads(r1.stdout.decode().strip().splitlines()[-1])
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(port), *base[1:], "--gpus", "2", "--single-device"],
                        cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert r2.returncode == 0, r2.stderr.decode()[-3000:]
    two = json.loads(r2.stdout.decode().strip().splitlines()[-1])
    assert two["n_gpus"] == 2 and two["value"] > 0 and two["rk4"]["error"] is None
    assert two["config"]["halo_wire_check"].endswith("ok on every rank")
    assert two["rk4"]["overlap_check"]["overlapped_equals_sequential"] is True
    assert two["rk4"]["state_checksums_after_2_steps"] == one["rk4"]["state_checksums_after_2_steps"]
    # the evaluation with the exchange of its inputs in front: measured, and not faster than the evaluation alone
    x = two["rhs_with_halo_exchange"]
    assert "error" not in x and x["ms_per_step"] > 0 and x["value"] > 0
    assert one["rhs_with_halo_exchange"] is None
    assert two["config"]["launcher"].startswith("torch.distributed.run")
    # the same run started PLAINLY (`python bench.py --gpus 2`, what the 1-GPU bench invocation looks like): bench.py
    # launches its own ranks as fresh child processes; the environment carries nothing of a launcher
    env = {k: v for k, v in os.environ.items()
           if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT",
                        "GROUP_RANK", "ROLE_RANK", "TORCHELASTIC_RUN_ID", "OMEGA_BENCH_LAUNCHER")}
    r3 = subprocess.run([*base, "--gpus", "2", "--single-device"], cwd=ROOT, env=env, stdout=subprocess.PIPE,
                        stderr=subprocess.PIPE, timeout=600)
    assert r3.returncode == 0, r3.stderr.decode()[-3000:]
    out_lines = r3.stdout.decode().strip().splitlines()
    assert len(out_lines) == 1, out_lines          # stdout carries exactly the one record
    plain = json.loads(out_lines[0])
    assert plain["config"]["launcher"].startswith("bench.py itself")
    assert plain["n_gpus"] == 2 and plain["value"] > 0 and plain["rk4"]["error"] is None
    assert plain["config"]["halo_wire_check"].endswith("ok on every rank")
    assert plain["rk4"]["overlap_check"]["overlapped_equals_sequential"] is True
    assert plain["rk4"]["state_checksums_after_2_steps"] == one["rk4"]["state_checksums_after_2_steps"]
    assert "error" not in plain["rhs_with_halo_exchange"]
    for key in ("cells", "levels", "tracers", "partition", "halo_width", "kernel_paths", "mesh_order"):
        assert plain["config"][key] == two["config"][key], key
