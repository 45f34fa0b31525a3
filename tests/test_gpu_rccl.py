"""GPU, two or more devices: the exchange executor against the REAL librccl -- one process per GPU, each started before
any GPU call, neighbour send/recv over xGMI between real devices, against the whole-frame oracle.

The pool's test box has ONE GPU, where this module skips (RCCL refuses two ranks on one device; the same executor is
run there through tests/native/fake_rccl.cpp by tests/test_gpu_exchange.py).  On a multi-GPU node it is the first thing
that proves the halo exchange end to end; bench.py --gpus N exercises the same path."""
import os
import subprocess
import sys

import numpy as np
import pytest

import reforge_amd as rf
from oracle import pixel
from tests import util

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def gpu_count():
    # counted in a child process: this process must not initialise devices it does not use
    try:
        out = subprocess.check_output([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], stderr=subprocess.DEVNULL, timeout=300)
        return int(out.decode().strip().splitlines()[-1])
    except Exception:
        return 0


def run_ranks(tmp_path, text, world, W, H, fmt, flags, seed, frames=1, source="fill"):
    cfg = tmp_path / "graph.cfg"
    cfg.write_text(text)
    env = dict(os.environ, RF_TEST_ONE_GPU_PER_RANK="1", RF_XCHG_TIMEOUT_S="60", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "exchange_worker.py"), str(r), str(world), str(W), str(H), str(fmt),
                               str(flags), str(seed), str(cfg), str(tmp_path), str(frames), source], env=env, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    errs = []
    for p in procs:
        try:
            _, err = p.communicate(timeout=300)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung in the RCCL exchange")
        errs.append((p.returncode, err[-1500:]))
    assert all(rc == 0 for rc, _ in errs), errs
    return np.concatenate([np.load(tmp_path / ("strip%d.npy" % r)) for r in range(world)], axis=0)


@pytest.mark.parametrize("text,flags", [(util.CHAIN3, 0), (util.CHAIN5, 0), (util.CHAIN5, rf.RF_GRAPH_NO_FUSION), (util.DIAMOND, 0)])
def test_real_rccl_exchange_reproduces_the_full_frame(tmp_path, text, flags):
    n = gpu_count()
    if n < 2:
        pytest.skip("needs two GPUs (this box has %d): RCCL refuses two ranks on one device" % n)
    world = min(n, 4)
    W, H, seed = 1000, 64 * world + 37, 0x5EED0004
    for fmt in (util.F32, util.U8):
        sub = tmp_path / ("fmt%d" % fmt)
        sub.mkdir()
        got = run_ranks(sub, text, world, W, H, fmt, flags, seed, frames=2)
        want = util.run_oracle(text, pixel.fill_synthetic(W, H, fmt, seed))
        util.assert_same(got, want, "real RCCL, world=%d flags=%d fmt=%d" % (world, flags, fmt))
