"""The N > 1 path on real GPUs: world size 2 over RCCL ("nccl" backend), the in-place arena broadcast over xGMI, each rank
generating its shard with no further collective.  Needs two MI355X: self-skips on the 1-GPU test box (the
same worker runs over gloo on CPU ranks in tests/test_dist_gloo.py)."""
import os

import pytest
import torch

from test_dist_gloo import ROOT, _torchrun

pytestmark = pytest.mark.gpu


def test_broadcast_and_sharded_generate_world2_rccl(hip_lib):
    if torch.cuda.device_count() < 2:
        pytest.skip(f"{torch.cuda.device_count()} GPU visible: the RCCL test needs 2")
    r = _torchrun(2, [os.path.join(ROOT, "tests", "_dist_worker.py"), hip_lib, "nccl"], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0 and "DIST_OK 2" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_rccl_path_world1(hip_lib):
    """The same worker at world size 1 on the one GPU of the test box: process-group start-up over RCCL, the torch tensor that
    ALIASES the engine's hipMalloc'ed arena (__cuda_array_interface__, no staging copy) accepted by the collective, the sharded
    generate and the host-side gather -- everything of the N > 1 path except a second rank."""
    r = _torchrun(1, [os.path.join(ROOT, "tests", "_dist_worker.py"), hip_lib, "nccl"], {"HSA_ENABLE_IPC_MODE_LEGACY": "0"})
    assert r.returncode == 0 and "DIST_OK 1" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
