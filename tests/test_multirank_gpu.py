"""The sharded N>1 path with the HIP stepper on every rank: two ranks share cuda:0 (the 8-GPU RCCL run is the driver's),
islands partitioned by edyn_amd.parallel.ShardedWorld, state gathered every step, one re-partition when an island crosses
shards - the trajectory must equal the unsharded GPU world bit for bit."""
import os
import numpy as np
import pytest
import torch.multiprocessing as mp

from test_multirank_gloo import _sharded_worker, _unsharded_states

pytestmark = pytest.mark.gpu


def test_sharded_hip_worlds_match_the_unsharded_gpu_world(tmp_path):
    steps = 90
    out = str(tmp_path / "sharded_gpu.npz")
    port = 29500 + (os.getpid() % 2000) + 11
    mp.spawn(_sharded_worker, args=(2, port, steps, out, True), nprocs=2, join=True)
    got = np.load(out)
    ref = _unsharded_states(steps, True)
    assert int(got["reparts"]) >= 1
    assert np.array_equal(got["states"], ref)
