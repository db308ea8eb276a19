"""The C ABI's RCCL wrappers on one GPU (a communicator of one rank: RCCL refuses two ranks on one device):
the library finds RCCL at run time, creates a communicator and every collective moves the data it should.
With one rank a sum over ranks, an exchange with oneself, a scatter and a gather are all copies."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_collectives_with_one_rank():
    from dlrm_amd import comm
    cid = comm.unique_id()
    assert len(cid) == comm.ID_BYTES
    c = comm.Comm(cid, 0, 1)
    try:
        g = torch.Generator(device="cuda").manual_seed(1)
        a = torch.randn(100003, device="cuda", generator=g)
        want = a.clone()
        c.all_reduce_sum_(a)
        assert torch.equal(a, want)
        send = torch.randn(4096 + 8, device="cuda", generator=g)
        recv = torch.zeros(5000, device="cuda")
        c.all_to_all(recv, send, [4096], [4096])
        assert torch.equal(recv[:4096], send[:4096]) and float(recv[4096:].abs().sum()) == 0.0
        out = torch.zeros(777, device="cuda")
        c.reduce_scatter_sum(out, send[:777].contiguous())
        assert torch.equal(out, send[:777])
        out2 = torch.zeros(777, device="cuda")
        c.all_gather(out2, send[1:778].contiguous())
        assert torch.equal(out2, send[1:778])
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # the caller's stream is honoured
            b = torch.ones(1 << 20, device="cuda")
            c.all_reduce_sum_(b)
        side.synchronize()
        assert float(b.sum()) == float(1 << 20)
    finally:
        c.close()


def test_bad_arguments():
    from dlrm_amd import comm
    from dlrm_amd._capi import DlrmError
    with pytest.raises(DlrmError, match="rank"):
        comm.Comm(comm.unique_id(), 3, 2)
