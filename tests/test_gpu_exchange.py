"""GPU: the EXCHANGE half of the multi-rank executor (rf_graph.cpp: exchange_rows on the comm stream,
three-part interior/boundary launches, event edges) run for real by 2..4 processes sharing GPU 0.
RCCL refuses two ranks on one device, so the processes load tests/native/fake_rccl.cpp -- the RCCL entry
points librfhip dlopens, with the real signatures, over shared memory -- in its place.  Everything
else is the product.  (The real library's ABI is covered by rf_comm_selftest; the schedule by
tests/test_dist_gloo.py; real multi-GPU runs are the driver's.)"""
import json
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
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.fixture(scope="module")
def fake_rccl_dir(tmp_path_factory):
    d = tmp_path_factory.mktemp("fake_rccl")
    so = str(d / "librccl.so.1")
    subprocess.check_call([HIPCC, "-O2", "-std=c++17", "-shared", "-fPIC", "-x", "c++", os.path.join(ROOT, "tests", "native", "fake_rccl.cpp"),
                           "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "reforge_amd", "csrc"), "-L/opt/rocm/lib", "-lamdhip64", "-lrt", "-o", so],
                          stderr=subprocess.DEVNULL)
    return str(d)


def run_ranks(fake_dir, tmp_path, text, world, W, H, fmt, flags, seed, frames=1, source="fill"):
    cfg = tmp_path / "graph.cfg"
    cfg.write_text(text)
    env = dict(os.environ, LD_LIBRARY_PATH=fake_dir + ":" + os.environ.get("LD_LIBRARY_PATH", ""))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "exchange_worker.py"), str(r), str(world), str(W), str(H), str(fmt),
                               str(flags), str(seed), str(cfg), str(tmp_path), str(frames), source], env=env, stderr=subprocess.PIPE, text=True)
             for r in range(world)]
    errs = []
    for p in procs:
        try:
            _, err = p.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise AssertionError("a rank hung in the exchange")
        errs.append((p.returncode, err[-1500:]))
    assert all(rc == 0 for rc, _ in errs), errs
    return np.concatenate([np.load(tmp_path / ("strip%d.npy" % r)) for r in range(world)], axis=0)


CASES = [(util.CHAIN3, 2, 0), (util.CHAIN5, 3, 0), (util.CHAIN5, 2, rf.RF_GRAPH_NO_FUSION), (util.DIAMOND, 2, 0), (util.CHAIN5_SPLIT, 4, 0)]


@pytest.mark.parametrize("text,world,flags", CASES)
def test_exchange_mode_reproduces_the_full_frame(fake_rccl_dir, tmp_path, text, world, flags):
    W, H, seed = 333, 257, 0x5EED0004
    for fmt in (util.F32, util.U8):
        sub = tmp_path / ("fmt%d" % fmt)
        sub.mkdir()
        got = run_ranks(fake_rccl_dir, sub, text, world, W, H, fmt, flags, seed)
        want = util.run_oracle(text, pixel.fill_synthetic(W, H, fmt, seed))
        util.assert_same(got, want, "exchange mode, world=%d flags=%d fmt=%d" % (world, flags, fmt))


@pytest.mark.parametrize("seed", range(6))
def test_exchange_mode_random_graphs(fake_rccl_dir, tmp_path, seed):
    rng = np.random.RandomState(8000 + seed)
    text = (util.random_dag if seed & 1 else util.random_graph)(rng)
    world = int(rng.randint(2, 4))
    W, H = int(rng.randint(40, 500)), int(rng.randint(150, 400))
    flags = (0, rf.RF_GRAPH_NO_FUSION)[(seed >> 1) & 1]
    fmt = (util.F32, util.U8)[seed % 2]
    if rf.Plan(rf.Config(text), flags).halo_schedule(True)[3] > H // world:
        pytest.skip("strips shorter than the widest halo")
    got = run_ranks(fake_rccl_dir, tmp_path, text, world, W, H, fmt, flags, seed)
    want = util.run_oracle(text, pixel.fill_synthetic(W, H, fmt, seed))
    util.assert_same(got, want, "exchange mode, world=%d flags=%d\n%s" % (world, flags, text))


def test_exchange_mode_several_frames(fake_rccl_dir, tmp_path):
    """Three frames back to back: the ghost rows of frame n+1 must not be exchanged before frame n's
    boundary launches have read them (src_ready / halo_ready edges)."""
    W, H, seed = 500, 301, 11
    got = run_ranks(fake_rccl_dir, tmp_path, util.CHAIN5, 3, W, H, util.F32, 0, seed, frames=3)
    util.assert_same(got, util.run_oracle(util.CHAIN5, pixel.fill_synthetic(W, H, util.F32, seed)), "three frames")


def test_exchange_mode_thin_strips(fake_rccl_dir, tmp_path):
    """Strips of 15 rows under a 7-row halo: too thin for an interior (rows <= 4 r), so the exchange
    runs on the launch's own stream in front of one whole-strip launch."""
    W, H, seed = 211, 60, 5
    got = run_ranks(fake_rccl_dir, tmp_path, util.CHAIN5, 4, W, H, util.U8, 0, seed)
    util.assert_same(got, util.run_oracle(util.CHAIN5, pixel.fill_synthetic(W, H, util.U8, seed)), "thin strips")


@pytest.mark.parametrize("flags", [0, rf.RF_GRAPH_NO_HALO_XCHG])
def test_uploaded_strips(fake_rccl_dir, tmp_path, flags):
    """Each rank uploads ITS rows of a host frame.  Exchange mode trades ghost rows per launch as
    before; over-fetch mode needs the neighbours' rows of the INPUT once, at upload (the cumulative
    halo), and nothing afterwards."""
    W, H, seed = 280, 190, 21
    for fmt in (util.F32, util.U8):
        sub = tmp_path / ("fmt%d" % fmt)
        sub.mkdir()
        got = run_ranks(fake_rccl_dir, sub, util.CHAIN5, 3, W, H, fmt, flags, seed, source="upload")
        util.assert_same(got, util.run_oracle(util.CHAIN5, pixel.fill_synthetic(W, H, fmt, seed)), "uploaded strips flags=%d" % flags)


def test_srgb_strips(fake_rccl_dir, tmp_path):
    """The sRGB boundary on strips: every rank uploads and downloads its rows of an RGBA8 frame."""
    from oracle import graph as og
    W, H, seed = 200, 160, 31
    rgba = pixel.fill_synthetic(W, H, util.U8, seed)
    for fmt in (util.F32, util.U8):
        sub = tmp_path / ("fmt%d" % fmt)
        sub.mkdir()
        got = run_ranks(fake_rccl_dir, sub, util.CHAIN3, 2, W, H, fmt, 0, seed, source="srgb")
        ref = og.GraphOracle(util.CHAIN3, W, H, fmt)
        ref.upload_srgb8(rgba)
        ref.execute()
        assert got.tobytes() == ref.download_srgb8().tobytes()


def _bench_rehearsal(fake_dir, port, fail, expect_rc=0):
    """bench.py as the driver launches it for N = 2, rehearsed on one GPU: both ranks on device 0, gloo for the bench's own
    collectives, the halo exchange through the RCCL test double (RF_RCCL_LIBRARY: torch has mapped its own librccl.so.1)."""
    env = dict(os.environ, RF_RCCL_LIBRARY=os.path.join(fake_dir, "librccl.so.1"), MASTER_ADDR="127.0.0.1")
    if fail:
        env["FAKE_RCCL_FAIL"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--rehearse", "--halo", "both", "--skip-cpu-baseline"]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert (p.returncode == 0) if expect_rc == 0 else (p.returncode != 0), (p.returncode, p.stderr[-3000:])
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_two_ranks_times_both_halo_schedules(fake_rccl_dir):
    out = _bench_rehearsal(fake_rccl_dir, 29541, fail=False)
    assert out["n_gpus"] == 2 and set(out["halo"]) >= {"overfetch", "exchange", "value_is"}, out.get("halo")
    assert "rccl_error" not in out
    assert "fake" in out["rccl_library"] or "librccl.so.1" in out["rccl_library"]
    assert set(out["strong_16k"]) >= {"overfetch", "exchange"} and "error" not in out["strong_16k"]["exchange"]


def test_bench_survives_a_failing_exchange(fake_rccl_dir):
    """The first real RCCL exchange between GPUs happens on the driver's node: if it fails, every rank must drop the exchange
    leg together and the run must still print its line (the over-fetch number) -- marked `exchange_failed`, the error under
    halo.exchange -- and leave with a NON-ZERO status: the mode that was asked for did not work."""
    out = _bench_rehearsal(fake_rccl_dir, 29542, fail=True, expect_rc=3)
    assert out["n_gpus"] == 2 and out["halo"]["value_is"] == "overfetch"
    assert out["exchange_failed"] is True and out["halo"]["exchange"]["error"] == out["rccl_error"] and out["rccl_error"], out
    assert out["value"] > 0 and "overfetch" in out["strong_16k"]
