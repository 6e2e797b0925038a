"""The native RCCL wire (omega_amd/csrc/Rccl.cpp).  A one-GPU box cannot host two RCCL ranks (RCCL refuses two ranks
on one device), so what runs here is everything short of a second peer: the library links and loads librccl, a
communicator is created from a unique id (ncclCommInitRank), and a grouped ncclSend / ncclRecv to the rank itself
moves data between device buffers in stream order on a non-blocking stream -- the same RcclComm::exchange the Halo
calls with its neighbour list.  The multi-peer exchange lists, the job-table pack / unpack kernels and the
overlap logic around the wire are covered by the 2- and 4-rank tests (tests/test_00_multirank_gpu.py) over the
host-staged test wire."""
import numpy as np
import pytest

import omega_amd as oa


def test_library_links_rccl():
    import subprocess
    out = subprocess.run(["readelf", "-d", oa.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl.so" in out
    nm = subprocess.run(["nm", "-D", "--undefined-only", oa.LIB_PATH], capture_output=True, text=True).stdout
    for sym in ("ncclCommInitRank", "ncclSend", "ncclRecv", "ncclGroupStart", "ncclGroupEnd", "ncclGetUniqueId"):
        assert sym in nm, sym


@pytest.mark.gpu
def test_self_exchange_on_a_user_stream():
    assert oa.device_count() > 0
    oa.device_init(0)
    uid = oa.RcclComm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = oa.RcclComm(uid, 1, 0)
    info = comm.info()
    assert info["nranks"] == 1 and info["rank"] == 0 and info["version"] >= 20000 and info["exchanges"] == 0
    n = 1 << 20
    src = oa.DeviceBuffer(np.arange(n, dtype=np.float64))
    dst = oa.DeviceBuffer(np.zeros(n))
    st = oa.Stream()
    for rep in range(3):   # repeated grouped exchanges on the same communicator, two messages in one group
        half = n // 2 * 8
        comm.exchange([0, 0], [src.ptr, src.ptr + half], [half, half], [dst.ptr, dst.ptr + half], [half, half], stream=st)
        st.synchronize()
        assert np.array_equal(dst.to_host(), src.host)
    assert comm.info()["exchanges"] == 3
    with pytest.raises(oa.OmegaAmdError, match="bad peer"):
        comm.exchange([1], [src.ptr], [8], [dst.ptr], [8], stream=st)


@pytest.mark.gpu
@pytest.mark.parametrize("wire_first", [True, False])
def test_halo_and_peer_wire_may_be_destroyed_in_either_order(wire_first):
    """The binding Halo <-> PeerWire is kept from both ends (bench.py and the multi-rank workers close the wire
    while the Halo it serves is still alive; the Halo's destructor then wrote into the freed wire).  One rank: a wire
    connected to itself, bound, and both objects destroyed -- wire first, then Halo, and the other way round -- with the
    survivor used in between: a Halo whose wire is gone simply has no wire, a wire whose Halo is gone can serve another."""
    from omega_amd.meshgen import planar_hex
    assert oa.device_count() > 0
    oa.device_init(0)
    gm = oa.GlobalMesh(planar_hex(8, 8, 30e3))
    d = oa.Decomp(gm, 1, 0, 3)
    halo = oa.Halo(d)
    wire = oa.PeerWire(1, 0, 4096)
    wire.connect([wire.handle()])
    halo.use_peer(wire)
    halo2 = oa.Halo(d)
    with pytest.raises(oa.OmegaAmdError, match="already serves another Halo"):
        halo2.use_peer(wire)
    if wire_first:
        wire.close()                      # omg_peer_destroy: the Halo is told (PeerWire.cpp: ~PeerWire)
        halo.check()                      # no wire: nothing to report, and no dangling pointer is followed
        oa.lib().omg_halo_destroy(halo.h)
        halo.h = None
    else:
        oa.lib().omg_halo_destroy(halo.h)  # ~Halo releases the wire ...
        halo.h = None
        halo2.use_peer(wire)               # ... which can serve another Halo
        assert wire.info()["status"] == 0
        wire.close()
        halo2.check()
    oa.device_synchronize()
