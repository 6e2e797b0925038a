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
