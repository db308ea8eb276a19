"""world_size 2 on ONE GPU: the sharded train step of dlrm_amd.model with the HIP kernels, both ranks on
device 0, against the single-process oracle.  RCCL refuses two ranks on one device, so the exchanges go over
gloo through host memory; everything else -- shard descriptors, packing around the exchanges, streams of
the overlapped schedule, the recorded step cut at the collectives -- is the code an 8-GPU run executes."""
import pytest
import torch
import torch.multiprocessing as mp

from tests.test_distributed_cpu import _free_port, _worker

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("replay", [False, True], ids=["eager", "replay"])
@pytest.mark.parametrize("kind", ["dot", "cat", "dot-rw", "cat-rw"])
def test_sharded_step_two_ranks_one_gpu(kind, replay):
    world = 2
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, port, kind, ret, "cuda", replay), nprocs=world, join=True)
    assert dict(ret) == {0: True, 1: True}


@pytest.mark.parametrize("replay", [False, True], ids=["eager", "replay"])
@pytest.mark.parametrize("kind", ["dot", "dot-rw", "cat-rw"])
def test_sharded_schedule_on_rccl_one_rank(kind, replay):
    """The sharded schedule on RCCL itself: a process group of ONE rank (backend "nccl"), DLRM(force_exchange=True):
    pack -> all_to_all_single -> unpack, reduce_scatter_tensor / all_gather_into_tensor for the row-wise tables,
    all_reduce of the dense gradients, all on device buffers, the embedding side on its own stream, eager and
    replayed between the collectives, against the single-process oracle."""
    port = _free_port()
    mgr = mp.get_context("spawn").Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(1, port, kind, ret, "cuda", replay, "nccl", True), nprocs=1, join=True)
    assert dict(ret) == {0: True}
