"""One rank of tests/test_gpu_exchange.py: the product in EXCHANGE mode (per-launch neighbour exchange,
interior/boundary split) with the test double of librccl on the loader path.  Run as a script:
  exchange_worker.py <rank> <world> <W> <H> <fmt> <flags> <seed> <config file> <out dir> <frames> [fill|upload|srgb]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

import reforge_amd as rf  # noqa: E402


def main():
    rank, world, W, H, fmt, flags, seed = (int(a) for a in sys.argv[1:8])
    text = open(sys.argv[8]).read()
    out_dir, frames = sys.argv[9], int(sys.argv[10])
    id_path = os.path.join(out_dir, "unique_id.bin")
    if rank == 0:
        uid = rf.Context.unique_id()
        with open(id_path + ".tmp", "wb") as fh:
            fh.write(uid)
        os.rename(id_path + ".tmp", id_path)
    else:
        for _ in range(3000):
            if os.path.exists(id_path):
                break
            time.sleep(0.01)
        uid = open(id_path, "rb").read()
    # RF_TEST_ONE_GPU_PER_RANK=1: rank r on device r with the REAL librccl (tests/test_gpu_rccl.py, needs world GPUs);
    # otherwise every rank shares GPU 0 through the test double
    device = rank if os.environ.get("RF_TEST_ONE_GPU_PER_RANK") == "1" else 0
    ctx = rf.Context(device, rank, world, uid)
    source = sys.argv[11] if len(sys.argv) > 11 else "fill"
    if os.environ.get("RF_TEST_SHADER_PATH"):                        # filter types that are files (tests/test_gpu_glsl.py: .comp files on every rank)
        rf.set_shader_path(os.environ["RF_TEST_SHADER_PATH"])
        rf.set_type_lookup(os.environ.get("RF_TEST_FILES_FIRST") == "1")
    g = rf.Graph(ctx, rf.Config(text), W, H, fmt, flags=flags)       # exchange mode unless the caller set RF_GRAPH_NO_HALO_XCHG
    y0, y1 = g.strip
    if source == "fill":
        g.fill_synthetic(seed)
    else:
        from oracle import pixel                                     # the test's input generator (not the product's)
        if source == "upload":
            g.upload_raw(pixel.fill_synthetic(W, H, fmt, seed)[y0:y1])          # a rank uploads ITS rows of the frame
        else:
            g.upload_srgb8(pixel.fill_synthetic(W, H, rf.RF_FORMAT_RGBA8, seed)[y0:y1])
    for _ in range(frames):
        g.execute()
    g.wait()
    np.save(os.path.join(out_dir, "strip%d.npy" % rank), g.download_srgb8() if source == "srgb" else g.download_raw())
    g.close()
    ctx.close()


if __name__ == "__main__":
    main()
