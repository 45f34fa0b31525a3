"""The C++ CLI host (reforge_amd/reforge, counterpart of src/main.rs): argument handling and
the dependency-free PNG reader/writer on CPU; one end-to-end frame on the GPU."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

from oracle import graph as og
from oracle import pixel
from tests import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLI = os.path.join(ROOT, "reforge_amd", "reforge")
CLI_SRC = [os.path.join(ROOT, "reforge_amd", "csrc", "host", f) for f in ("reforge_main.cpp", "png_io.h")]


@pytest.fixture(autouse=True, scope="module")
def cli_is_fresh():
    """The binary is a build artefact (git-ignored): a missing one, or one older than its sources,
    must fail these tests instead of quietly exercising something stale."""
    assert os.path.exists(CLI), "reforge_amd/reforge is not built: run `python -c 'import __graft_entry__ as g; g.build()'`"
    newest = max(os.path.getmtime(f) for f in CLI_SRC)
    assert os.path.getmtime(CLI) >= newest, "reforge_amd/reforge is older than csrc/host/*: rebuild"



def write_png(path, img, filt):
    """Reference PNG encoder (zlib level 9, every scanline filter) for the decoder test."""
    h, w, c = img.shape
    raw = b""
    prev = np.zeros((w * c,), np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        ft = filt(y)
        a = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        cc = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
        if ft == 0:
            pred = 0
        elif ft == 1:
            pred = a
        elif ft == 2:
            pred = prev
        elif ft == 3:
            pred = (a + prev) >> 1
        else:
            p = a + prev - cc
            pa, pb, pc = abs(p - a), abs(p - prev), abs(p - cc)
            pred = np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, prev, cc))
        raw += bytes([ft]) + ((cur - pred) & 255).astype(np.uint8).tobytes()
        prev = cur

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    ct = {1: 0, 2: 4, 3: 2, 4: 6}[c]
    with open(path, "wb") as fh:
        fh.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, ct, 0, 0, 0)) +
                 chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def run_cli(*args):
    return subprocess.run([CLI] + list(args), capture_output=True, text=True)


@pytest.mark.parametrize("channels", [1, 2, 3, 4])
def test_png_decoder_and_writer(tmp_path, channels):
    rng = np.random.RandomState(channels)
    img = rng.randint(0, 256, (37, 53, channels)).astype(np.uint8)
    img[5:20] = 77                                       # long matches: dynamic Huffman + back-references
    src, raw, again = str(tmp_path / "in.png"), str(tmp_path / "out.rgba"), str(tmp_path / "again.png")
    write_png(src, img, lambda y: y % 5)
    assert run_cli("-i", src, "-o", raw, "--decode-only").returncode == 0
    got = np.fromfile(raw, np.uint8).reshape(37, 53, 4)
    want = np.zeros((37, 53, 4), np.uint8)
    if channels == 1:
        want[..., :3], want[..., 3] = img, 255
    elif channels == 2:
        want[..., :3], want[..., 3] = img[..., :1], img[..., 1]
    elif channels == 3:
        want[..., :3], want[..., 3] = img, 255
    else:
        want = img
    assert (got == want).all()
    # our own writer (stored deflate) is readable by zlib-based decoders and by ourselves
    assert run_cli("-i", src, "-o", again, "--decode-only").returncode == 0
    assert run_cli("-i", again, "-o", raw, "--decode-only").returncode == 0
    assert (np.fromfile(raw, np.uint8).reshape(37, 53, 4) == want).all()
    data = open(again, "rb").read()
    idat = data[data.index(b"IDAT") + 4:data.index(b"IEND") - 8]
    assert len(zlib.decompress(idat)) == 37 * (53 * 4 + 1)


def test_cli_argument_rules(tmp_path):
    """main.rs:76-83: headless needs -o; a config and a positional shader exclude each other."""
    r = run_cli()
    assert r.returncode != 0 and "pass -o" in r.stderr
    cfg = tmp_path / "p.cfg"
    cfg.write_text("input -> passthrough -> output")
    r = run_cli("--config", str(cfg), "sharpen", "-o", str(tmp_path / "o.png"))
    assert r.returncode == 1 and "Cannot specify both a config and shader file" in r.stderr
    r = run_cli("-o", str(tmp_path / "o.png"), "--shader-format", "rgba16")
    assert r.returncode == 2
    # the default graph reads 'input' (render.rs:115) but no input was given (config.rs:201)
    r = run_cli("-o", str(tmp_path / "o.png"))
    assert r.returncode == 1 and "no input image was specified" in r.stderr


@pytest.mark.gpu
def test_cli_end_to_end(tmp_path):
    """reforge -i in.png --config chain.cfg -o out.png == upload_srgb8 -> graph -> download_srgb8 of the oracle."""
    rgba = pixel.fill_synthetic(160, 90, util.U8, 77)
    rgba[..., 3] = 255
    src, dst, cfg = str(tmp_path / "in.png"), str(tmp_path / "out.rgba"), tmp_path / "chain.cfg"
    write_png(src, rgba, lambda y: 4)
    cfg.write_text(util.CHAIN3)
    for fmt_name, fmt in (("rgba32f", util.F32), ("rgba8", util.U8)):
        r = run_cli("-i", src, "--config", str(cfg), "-o", dst, "--shader-format", fmt_name)
        assert r.returncode == 0, r.stderr
        assert "GPU: {blur+grade+sharp: " in r.stderr          # the status line of main.rs:157
        ref = og.GraphOracle(util.CHAIN3, 160, 90, fmt)
        ref.upload_srgb8(rgba)
        ref.execute()
        assert np.fromfile(dst, np.uint8).reshape(90, 160, 4).tobytes() == ref.download_srgb8().tobytes()
    # single-shader mode (config.rs:77-90)
    r = run_cli("-i", src, "sharpen.comp", "-o", dst, "--no-fusion")
    assert r.returncode == 0 and "GPU: {sharpen: " in r.stderr


@pytest.mark.gpu
def test_cli_live_reload(tmp_path):
    """--frames N --watch is the windowed loop of main.rs:134-182 without the window: the config's
    mtime is polled every frame (render.rs:138-165); a config that does not parse leaves the old
    graph running (render.rs:121-136), a good one replaces it, and the last frame is what -o gets."""
    import time
    rgba = pixel.fill_synthetic(96, 64, util.U8, 78)
    rgba[..., 3] = 255
    src, dst, cfg, log = str(tmp_path / "in.png"), str(tmp_path / "out.rgba"), tmp_path / "live.cfg", tmp_path / "err.log"
    write_png(src, rgba, lambda y: 0)
    cfg.write_text("input -> passthrough -> output")
    t0 = int(os.path.getmtime(cfg))

    def wait_for(text, seconds=60.0):
        end = time.time() + seconds
        while time.time() < end:
            if text in log.read_text(errors="replace"):
                return True
            time.sleep(0.01)
        return False

    with open(log, "wb") as err:
        p = subprocess.Popen([CLI, "-i", src, "--config", str(cfg), "-o", dst, "--frames", "160", "--watch",
                              "--frame-interval-ms", "25"], stderr=err)
        try:
            assert wait_for("GPU: {passthrough: ")                 # the loop runs and prints main.rs:157's line
            cfg.write_text("input -> -> output")
            os.utime(cfg, (t0 + 2, t0 + 2))                        # mtime has whole-second resolution (utils.rs:33-42)
            assert wait_for("Unrecognized token")
            assert p.poll() is None                                # still rendering with the old graph
            cfg.write_text("input -> sharp -> output\nsharp: sharpen { amount: 0.75 }")
            os.utime(cfg, (t0 + 4, t0 + 4))
            assert wait_for("GPU: {sharp: ")
            assert p.wait(timeout=120) == 0
        finally:
            if p.poll() is None:
                p.kill()
    ref = og.GraphOracle("input -> sharp -> output\nsharp: sharpen { amount: 0.75 }", 96, 64, util.F32)
    ref.upload_srgb8(rgba)
    ref.execute()
    assert np.fromfile(dst, np.uint8).reshape(64, 96, 4).tobytes() == ref.download_srgb8().tobytes()


@pytest.mark.gpu
def test_cli_user_stage_files_and_their_live_reload(tmp_path):
    """--shader-path: a node type the registry lacks is the file {shader_path}/{type}.stage.hip (config.rs:59-75), compiled when
    the graph is built; --watch polls the stage files of the running graph like reload_changed_pipelines (render.rs:225-249):
    an edited file rebuilds the graph, one that no longer compiles prints the compiler's message and keeps the old graph."""
    import shutil
    import time
    import reforge_amd as rf
    rgba = pixel.fill_synthetic(96, 64, util.U8, 79)
    rgba[..., 3] = 255
    shaders = tmp_path / "sh"
    shaders.mkdir()
    for f in ("invert.stage.hip", "edge_detect.stage.hip"):
        shutil.copy(os.path.join(ROOT, "shaders", f), shaders / f)
    src, dst, cfg, log = str(tmp_path / "in.png"), str(tmp_path / "out.rgba"), tmp_path / "live.cfg", tmp_path / "err.log"
    write_png(src, rgba, lambda y: 0)
    cfg.write_text("input -> neg -> output\nneg: invert { enabled: true, strength: 1.0 }")
    # one frame: equals the Python host on the same graph (same library, same stage file)
    r = run_cli("-i", src, "--config", str(cfg), "-o", dst, "--shader-path", str(shaders))
    assert r.returncode == 0 and "GPU: {neg: " in r.stderr, r.stderr
    old = rf.shader_path()
    rf.set_shader_path(str(shaders))
    try:
        ctx = rf.Context(0)
        g = rf.Graph(ctx, rf.Config(cfg.read_text()), 96, 64, rf.RF_FORMAT_RGBA32F)
        g.upload_srgb8(rgba)
        g.execute(); g.wait()
        want = g.download_srgb8()
        g.close()
        ctx.close()
    finally:
        rf.set_shader_path(old)
    assert np.fromfile(dst, np.uint8).reshape(64, 96, 4).tobytes() == want.tobytes()
    assert not np.array_equal(want[..., :3], rgba[..., :3])
    # without --shader-path pointing at the file the type does not exist (Shader::from_path -> None, utils.rs:23)
    r = run_cli("-i", src, "--config", str(cfg), "-o", dst, "--shader-path", str(tmp_path / "nowhere"))
    assert r.returncode != 0 and "no such filter" in r.stderr

    def wait_for(text, seconds=90.0):
        end = time.time() + seconds
        while time.time() < end:
            if text in log.read_text(errors="replace"):
                return True
            time.sleep(0.01)
        return False

    stage = shaders / "invert.stage.hip"
    good = stage.read_text()
    with open(log, "wb") as err:
        p = subprocess.Popen([CLI, "-i", src, "--config", str(cfg), "-o", dst, "--shader-path", str(shaders), "--frames", "400", "--watch",
                              "--frame-interval-ms", "25"], stderr=err)
        try:
            assert wait_for("GPU: {neg: ")
            stage.write_text(good.replace("(1.0f - c.x) - c.x", "(1.0f - c.x) - undeclared_thing"))       # no longer compiles
            assert wait_for("undeclared_thing")
            assert p.poll() is None                                # still rendering with the old graph
            stage.write_text(good.replace("if (!p.enabled) return c;", "return c;"))                       # now a passthrough
            time.sleep(0.05)
            os.utime(stage, None)
            time.sleep(1.5)                                        # several frames with the rebuilt graph
            assert p.poll() is None or p.returncode == 0
            assert p.wait(timeout=180) == 0
        finally:
            if p.poll() is None:
                p.kill()
    # the last frame came from the edited stage: a passthrough through the sRGB boundary = the input codes
    assert np.fromfile(dst, np.uint8).reshape(64, 96, 4).tobytes() == rgba.tobytes()


@pytest.mark.gpu
def test_cli_runs_a_directory_of_glsl_shaders_as_it_is(tmp_path):
    """the reference's plugin contract end to end: --shader-path holds .comp files (GLSL 450 compute, config.rs:59-75), the config
    names them, --shader-files-first lets them stand for built-in names too; an edit that no longer translates keeps the old graph"""
    import shutil
    import time
    import reforge_amd as rf
    rgba = pixel.fill_synthetic(96, 64, util.U8, 83)
    rgba[..., 3] = 255
    shaders = tmp_path / "sh"
    shaders.mkdir()
    for f in ("invert.comp", "sharpen.comp"):
        shutil.copy(os.path.join(ROOT, "shaders", f), shaders / f)
    src, dst, cfg, log = str(tmp_path / "in.png"), str(tmp_path / "out.rgba"), tmp_path / "live.cfg", tmp_path / "err.log"
    write_png(src, rgba, lambda y: 0)
    cfg.write_text("input -> neg -> sh -> output\nneg: invert { enabled: true, strength: 1.0 }\nsh: sharpen { amount: 0.5 }")
    r = run_cli("-i", src, "--config", str(cfg), "-o", dst, "--shader-path", str(shaders), "--shader-files-first")
    assert r.returncode == 0 and "GPU: {neg+sh: " in r.stderr, r.stderr      # both types are files -- a point shader and a 3 x 3 stencil: row stages, ONE launch
    got = np.fromfile(dst, np.uint8).reshape(64, 96, 4).copy()
    r = run_cli("-i", src, "--config", str(cfg), "-o", dst, "--shader-path", str(shaders))      # default lookup: sharpen is the built-in kernel, invert the file
    assert r.returncode == 0, r.stderr
    assert np.fromfile(dst, np.uint8).reshape(64, 96, 4).tobytes() == got.tobytes()            # the same bits either way
    # `reforge [shader]`: the path of a shader FILE is the whole pipeline (single_shader_parse, config.rs:77-90), wherever it lies
    one = str(tmp_path / "one.rgba")
    r = run_cli("-i", src, "-o", one, str(shaders / "sharpen.comp"))
    assert r.returncode == 0 and "GPU: {sharpen: " in r.stderr, r.stderr
    (tmp_path / "only.cfg").write_text("input -> sharpen -> output")      # amount = 0 either way (absent parameters are zero, render.rs:200-203)
    r = run_cli("-i", src, "--config", str(tmp_path / "only.cfg"), "-o", dst)
    assert r.returncode == 0, r.stderr
    assert np.fromfile(one, np.uint8).tobytes() == np.fromfile(dst, np.uint8).tobytes()
    with open(log, "wb") as err:
        p = subprocess.Popen([CLI, "-i", src, "--config", str(cfg), "-o", dst, "--shader-path", str(shaders), "--shader-files-first", "--frames", "300", "--watch",
                              "--frame-interval-ms", "25"], stderr=err)
        try:
            end = time.time() + 90
            while time.time() < end and "GPU: {neg+sh: " not in log.read_text(errors="replace"):
                time.sleep(0.01)
            good = (shaders / "invert.comp").read_text()
            (shaders / "invert.comp").write_text(good.replace("uniform Params", "uniform samplerCube nope; uniform Params"))
            end = time.time() + 90
            while time.time() < end and "invert.comp:" not in log.read_text(errors="replace"):
                time.sleep(0.01)
            assert "invert.comp:" in log.read_text(errors="replace") and p.poll() is None      # refused with file:line, still rendering
            (shaders / "invert.comp").write_text(good)
            assert p.wait(timeout=180) == 0
        finally:
            if p.poll() is None:
                p.kill()
    assert np.fromfile(dst, np.uint8).reshape(64, 96, 4).tobytes() == got.tobytes()
