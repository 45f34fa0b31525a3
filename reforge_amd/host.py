"""Python host over the C ABI: the same objects and call order the reference's
`Render` drives (src/render.rs, src/main.rs:134-182), with numpy arrays standing in
for the mapped staging buffer.  No compute happens here: every method is one call
into librfhip.so.
"""
import ctypes as C
import os
from dataclasses import dataclass
from typing import Optional

import numpy as np

from . import _lib
from ._lib import lib

FILE_INPUT = "rf:file-input"      # src/vulkan/pipeline_graph.rs:22
FINAL_OUTPUT = "rf:final-output"  # src/vulkan/pipeline_graph.rs:23


class RfError(RuntimeError):
    """A non-zero rf_status; .status holds the code, the text is rf_last_error()."""

    def __init__(self, status, where):
        self.status = status
        msg = lib().rf_last_error().decode("utf-8", "replace")
        super().__init__("%s failed (rf_status %d): %s" % (where, status, msg))


def _check(status, where):
    if status != _lib.RF_OK:
        raise RfError(status, where)


def _s(b):
    return None if b is None else b.decode("utf-8")


def get_dim(width, height, new_width=None, new_height=None):
    """utils::get_dim, src/utils.rs:56-74: aspect-preserving resize arithmetic in f32."""
    if new_width is not None and new_height is not None:
        return new_width, new_height
    w, h = width, height
    f32 = np.float32
    if new_width is not None:
        w = new_width
        h = int(f32(f32(w) / f32(width)) * f32(height))
    elif new_height is not None:
        h = new_height
        w = int(f32(f32(h) / f32(height)) * f32(width))
    return w, h


def registry_types():
    L = lib()
    return [_s(L.rf_registry_type_name(i)) for i in range(L.rf_registry_num_types())]


def registry_binding(type_name, descriptor):
    return lib().rf_registry_binding(type_name.encode(), descriptor.encode())


def config_syntax(text):
    """The syntax tree of a config text ({"exprs": [...]}, see rf_config_syntax in include/rfhip.h) -- the generated parser
    of src/config/config.rs:105 alone; raises RfError (status 2) for a text the grammar rejects."""
    import json
    if "\x00" in text:      # the C ABI takes NUL-terminated text; the grammar has no token for a NUL (the reference: InvalidToken)
        raise ValueError("a config text cannot hold a NUL character")
    raw = text.encode("utf-8")
    cap = 4 * len(raw) + 256
    while True:
        buf = C.create_string_buffer(cap)
        n = C.c_size_t()
        st = lib().rf_config_syntax(raw, buf, cap, C.byref(n))
        if st == 1 and n.value + 1 > cap:
            cap = n.value + 1
            continue
        _check(st, "rf_config_syntax")
        return json.loads(buf.raw[:n.value].decode("utf-8"))


def set_shader_path(path):
    """Directory searched for {type}.stage.hip when a config names a type the built-in registry lacks (config.rs:59-75)."""
    _check(lib().rf_set_shader_path(os.fsencode(path) if path else b""), "rf_set_shader_path")


def shader_path():
    return lib().rf_shader_path().decode()


def _glsl_text(fn, what, type_name, text):
    raw = text.encode("utf-8")
    cap = 16 * len(raw) + (1 << 16)
    while True:
        buf = C.create_string_buffer(cap)
        n = C.c_size_t()
        st = fn(type_name.encode(), raw, buf, cap, C.byref(n))
        if st == 1 and n.value + 1 > cap:
            cap = n.value + 1
            continue
        _check(st, what)
        return buf.raw[:n.value].decode("utf-8")


def glsl_translate(type_name, text):
    """HIP device source of the GLSL compute shader `text` ({shader_path}/{type_name}.comp, rf_glsl_translate); raises RfError
    with "type.comp:LINE: why" for a file outside the translated subset."""
    return _glsl_text(lib().rf_glsl_translate, "rf_glsl_translate", type_name, text)


def glsl_reflect(type_name, text):
    """What spirv-reflect gives the reference for the shader (src/vulkan/shader.rs:106-160), as a dict (rf_glsl_reflect)."""
    import json
    return json.loads(_glsl_text(lib().rf_glsl_reflect, "rf_glsl_reflect", type_name, text))


def set_type_lookup(files_first):
    """True: a file in the shader path wins over a built-in type of the same name (the reference's rule); False (default): the built-in."""
    _check(lib().rf_set_type_lookup(1 if files_first else 0), "rf_set_type_lookup")


def strip_rows(height, world, rank):
    y0, y1 = C.c_int(), C.c_int()
    _check(lib().rf_strip_rows(height, world, rank, C.byref(y0), C.byref(y1)), "rf_strip_rows")
    return y0.value, y1.value


def comm_selftest(device=0, nbytes=1 << 20):
    """One-rank RCCL send/recv round trip on `device` (rf_comm_selftest)."""
    _check(lib().rf_comm_selftest(device, nbytes), "rf_comm_selftest")


class Config:
    """config::Config (src/config/config.rs:35-38) parsed by the library."""

    def __init__(self, text=None, expects_input=True, single=None):
        self._h = C.c_void_p()
        if single is not None:
            _check(lib().rf_config_single(single.encode(), int(expects_input), C.byref(self._h)), "rf_config_single")
        else:
            _check(lib().rf_config_parse(text.encode(), int(expects_input), C.byref(self._h)), "rf_config_parse")

    def __del__(self):
        if getattr(self, "_h", None):
            lib().rf_config_destroy(self._h)
            self._h = None

    @property
    def handle(self):
        return self._h

    def nodes(self):
        """name -> dict(type, inputs[(resource, descriptor)], outputs[...], params{})"""
        L, h, out = lib(), self._h, {}
        for n in range(L.rf_config_num_nodes(h)):
            out[_s(L.rf_config_node_name(h, n))] = {
                "type": _s(L.rf_config_node_type(h, n)),
                "inputs": [(_s(L.rf_config_node_input_resource(h, n, i)), _s(L.rf_config_node_input_descriptor(h, n, i)))
                           for i in range(L.rf_config_node_num_inputs(h, n))],
                "outputs": [(_s(L.rf_config_node_output_resource(h, n, i)), _s(L.rf_config_node_output_descriptor(h, n, i)))
                            for i in range(L.rf_config_node_num_outputs(h, n))],
                "params": {_s(L.rf_config_node_param_key(h, n, i)): _s(L.rf_config_node_param_value(h, n, i))
                           for i in range(L.rf_config_node_num_params(h, n))},
            }
        return out


class Plan:
    """Layers + image aliasing (src/vulkan/pipeline_graph.rs:358-497) of a Config."""

    def __init__(self, config=None, flags=0, _borrowed=None):
        self._own = _borrowed is None
        if _borrowed is not None:
            self._h = C.c_void_p(_borrowed)
        else:
            self._h = C.c_void_p()
            _check(lib().rf_plan_create(config.handle, flags, C.byref(self._h)), "rf_plan_create")

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "_h", None):
            lib().rf_plan_destroy(self._h)
            self._h = None

    def layers(self):
        L, h = lib(), self._h
        return [[_s(L.rf_plan_layer_node(h, l, i)) for i in range(L.rf_plan_layer_size(h, l))]
                for l in range(L.rf_plan_num_layers(h))]

    def aliases(self):
        L, h = lib(), self._h
        return {_s(L.rf_plan_alias_from(h, i)): _s(L.rf_plan_alias_to(h, i)) for i in range(L.rf_plan_num_aliases(h))}

    def images(self):
        L, h = lib(), self._h
        return [_s(L.rf_plan_image_name(h, i)) for i in range(L.rf_plan_num_images(h))]

    def resolve(self, resource):
        return _s(lib().rf_plan_resolve(self._h, resource.encode()))

    def buffers(self):
        """{allocated storage-buffer name: bytes} (SSBO edges, pipeline_graph.rs:142-175)."""
        L, h = lib(), self._h
        return {_s(L.rf_plan_buffer_name(h, i)): L.rf_plan_buffer_bytes(h, i) for i in range(L.rf_plan_num_buffers(h))}

    def resolve_buffer(self, resource):
        return _s(lib().rf_plan_resolve_buffer(self._h, resource.encode()))

    def launches(self):
        L, h = lib(), self._h
        return [_s(L.rf_plan_launch_label(h, i)) for i in range(L.rf_plan_num_launches(h))]

    def launch_info(self):
        """One dict per kernel launch, in execution order."""
        L, h = lib(), self._h
        return [{"label": _s(L.rf_plan_launch_label(h, i)),
                 "layer": L.rf_plan_launch_layer(h, i),
                 "members": [_s(L.rf_plan_launch_member(h, i, k)) for k in range(L.rf_plan_launch_num_members(h, i))],
                 "member_slots": [L.rf_plan_launch_member_slot(h, i, k) for k in range(L.rf_plan_launch_num_members(h, i))],
                 "inputs": [_s(L.rf_plan_launch_input(h, i, k)) for k in range(L.rf_plan_launch_num_inputs(h, i))],
                 "output": _s(L.rf_plan_launch_output(h, i)),
                 "outputs": [_s(L.rf_plan_launch_output_at(h, i, k)) for k in range(L.rf_plan_launch_num_outputs(h, i))],
                 "radius": L.rf_plan_launch_radius(h, i),
                 "serial": bool(L.rf_plan_launch_serial(h, i))} for i in range(L.rf_plan_num_launches(h))]

    def signature(self):
        """64-bit digest of the launch list: row-strip ranks in exchange mode must agree on it."""
        return int(lib().rf_plan_signature(self._h))

    def needs_jit(self):
        """Per launch: True if its kernel is not in the ahead-of-time catalogue (compiled at graph creation)."""
        L, h = lib(), self._h
        return [bool(L.rf_plan_launch_needs_jit(h, i)) for i in range(L.rf_plan_num_launches(h))]

    def jit_compile(self, fmt=_lib.RF_FORMAT_RGBA32F, texels_per_lane=1):
        """Compile every such kernel for gfx950 without a device; returns the total code size in bytes."""
        n = C.c_size_t()
        _check(lib().rf_plan_jit_compile_texels(self._h, fmt, texels_per_lane, C.byref(n)), "rf_plan_jit_compile")
        return n.value

    def halo_schedule(self, exchange=True):
        """(need_src[], need_dst[], need_input, ghost) of a row-strip partition."""
        n = lib().rf_plan_num_launches(self._h)
        ns, nd = (C.c_int * max(n, 1))(), (C.c_int * max(n, 1))()
        ni, gh = C.c_int(), C.c_int()
        _check(lib().rf_plan_halo_schedule(self._h, int(exchange), ns, nd, n, C.byref(ni), C.byref(gh)), "rf_plan_halo_schedule")
        return list(ns[:n]), list(nd[:n]), ni.value, gh.value


class Context:
    """VkCore (src/vulkan/core.rs:66-146): one GPU, optionally one rank of a node-wide job."""

    def __init__(self, device=0, rank=0, world=1, unique_id=None):
        self._h = C.c_void_p()
        if world > 1:
            # unique_id None: a rank without a communicator (over-fetch graphs only)
            buf = C.create_string_buffer(bytes(unique_id), 128) if unique_id is not None else None
            _check(lib().rf_ctx_create_dist(device, rank, world, buf, C.byref(self._h)), "rf_ctx_create_dist")
        else:
            _check(lib().rf_ctx_create(device, C.byref(self._h)), "rf_ctx_create")

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        _check(lib().rf_comm_unique_id(buf), "rf_comm_unique_id")
        return buf.raw

    def close(self):
        if getattr(self, "_h", None):
            lib().rf_ctx_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def handle(self):
        return self._h

    @property
    def rank(self):
        return lib().rf_ctx_rank(self._h)

    @property
    def world(self):
        return lib().rf_ctx_world(self._h)

    @property
    def arch(self):
        return _s(lib().rf_ctx_device_arch(self._h))

    def synchronize(self):
        _check(lib().rf_ctx_synchronize(self._h), "rf_ctx_synchronize")

    def copy_bandwidth(self, nbytes=1 << 30, iters=20):
        g = C.c_float()
        _check(lib().rf_ctx_copy_bandwidth(self._h, nbytes, iters, C.byref(g)), "rf_ctx_copy_bandwidth")
        return g.value


_DTYPES = {_lib.RF_FORMAT_RGBA8: np.uint8, _lib.RF_FORMAT_RGBA32F: np.float32}


class Graph:
    """PipelineGraph + its per-frame resources (src/vulkan/pipeline_graph.rs:43-57)."""

    def __init__(self, ctx, config, width, height, fmt=_lib.RF_FORMAT_RGBA32F, num_frames=1, flags=0,
                 rows_per_chunk=0, conv_path=0, exec_flags=0, texels_per_lane=0):
        self.ctx, self.width, self.height, self.format = ctx, width, height, fmt
        self._h = C.c_void_p()
        opt = _lib.GraphOptions(width, height, fmt, num_frames, flags, rows_per_chunk or 0, conv_path, exec_flags, texels_per_lane)
        _check(lib().rf_graph_create(ctx.handle, config.handle, C.byref(opt), C.byref(self._h)), "rf_graph_create")
        y0, y1 = C.c_int(), C.c_int()
        _check(lib().rf_graph_strip(self._h, C.byref(y0), C.byref(y1)), "rf_graph_strip")
        self.strip = (y0.value, y1.value)
        self.rows = y1.value - y0.value

    def close(self):
        if getattr(self, "_h", None):
            lib().rf_graph_destroy(self._h)
            self._h = None

    __del__ = close

    @property
    def plan(self):
        return Plan(_borrowed=lib().rf_graph_plan(self._h))

    # -- parameters (render.rs:167-223) ------------------------------------------
    def set_param(self, node, name, value):
        if isinstance(value, bool):
            v, t = C.c_int32(int(value)), _lib.RF_PARAM_BOOL
        elif isinstance(value, (int, np.integer)):
            v, t = C.c_int32(int(value)), _lib.RF_PARAM_I32
        else:
            v, t = C.c_float(float(value)), _lib.RF_PARAM_F32
        _check(lib().rf_graph_set_param(self._h, node.encode(), name.encode(), t, C.byref(v)), "rf_graph_set_param")

    def set_weights(self, node, weights):
        w = np.ascontiguousarray(weights, np.float32)
        _check(lib().rf_graph_set_weights(self._h, node.encode(), w.ctypes.data_as(C.POINTER(C.c_float)), w.size),
               "rf_graph_set_weights")

    def set_time(self, seconds):
        _check(lib().rf_graph_set_time(self._h, float(seconds)), "rf_graph_set_time")

    # -- input (render.rs:264-313) -------------------------------------------------
    def _texel_array(self, arr, dtype):
        a = np.ascontiguousarray(arr, dtype)
        if a.shape != (self.rows, self.width, 4):
            raise ValueError("expected shape %s, got %s" % ((self.rows, self.width, 4), a.shape))
        return a

    def upload_raw(self, texels):
        a = self._texel_array(texels, _DTYPES[self.format])
        _check(lib().rf_graph_upload_raw(self._h, a.ctypes.data, a.strides[0]), "rf_graph_upload_raw")

    def upload_srgb8(self, rgba):
        a = self._texel_array(rgba, np.uint8)
        _check(lib().rf_graph_upload_srgb8(self._h, a.ctypes.data, a.strides[0]), "rf_graph_upload_srgb8")

    def fill_synthetic(self, seed):
        _check(lib().rf_graph_fill_synthetic(self._h, seed & 0xFFFFFFFF), "rf_graph_fill_synthetic")

    def fill_structured(self):
        _check(lib().rf_graph_fill_structured(self._h), "rf_graph_fill_structured")

    # -- execution (render.rs:359-404,:441-495 -> command.rs:166-242) ---------------
    def execute(self, slot=0):
        _check(lib().rf_graph_execute(self._h, slot), "rf_graph_execute")

    def wait(self, slot=0):
        _check(lib().rf_graph_wait(self._h, slot), "rf_graph_wait")

    # -- output (render.rs:406-433) --------------------------------------------------
    def download_raw(self, slot=0):
        out = np.empty((self.rows, self.width, 4), _DTYPES[self.format])
        _check(lib().rf_graph_download_raw(self._h, slot, out.ctypes.data, out.strides[0]), "rf_graph_download_raw")
        return out

    def download_srgb8(self, slot=0):
        out = np.empty((self.rows, self.width, 4), np.uint8)
        _check(lib().rf_graph_download_srgb8(self._h, slot, out.ctypes.data, out.strides[0]), "rf_graph_download_srgb8")
        return out

    def download_rows(self, y0, y1, slot=0):
        """Rows [y0, y1) of the strip's output image (raw texels)."""
        out = np.empty((y1 - y0, self.width, 4), _DTYPES[self.format])
        _check(lib().rf_graph_download_rows(self._h, slot, y0, y1, out.ctypes.data, out.strides[0]), "rf_graph_download_rows")
        return out

    def download_image(self, resource, slot=0):
        out = np.empty((self.rows, self.width, 4), _DTYPES[self.format])
        _check(lib().rf_graph_download_image(self._h, slot, resource.encode(), out.ctypes.data, out.strides[0]),
               "rf_graph_download_image")
        return out

    # -- GpuTimer (vkutils.rs:47-135) ------------------------------------------------
    def node_times(self, slot=0):
        cap = 256
        names = (C.c_char_p * cap)()
        ms = (C.c_float * cap)()
        n = C.c_int(cap)
        _check(lib().rf_graph_node_times(self._h, slot, names, ms, C.byref(n)), "rf_graph_node_times")
        return [(_s(names[i]), ms[i]) for i in range(n.value)]

    def times_string(self, slot=0):
        buf = C.create_string_buffer(4096)
        _check(lib().rf_graph_times_string(self._h, slot, buf, len(buf)), "rf_graph_times_string")
        return buf.value.decode()

    # -- measurement helpers ----------------------------------------------------------
    def time_frames(self, iters):
        ms = C.c_float()
        _check(lib().rf_graph_time_frames(self._h, iters, C.byref(ms)), "rf_graph_time_frames")
        return ms.value

    def time_frames_rotating(self, iters):
        """Total GPU ms of `iters` frames, frame i on slot i % num_frames, all on one queue (the cache-cold rate)."""
        ms = C.c_float()
        _check(lib().rf_graph_time_frames_rotating(self._h, iters, C.byref(ms)), "rf_graph_time_frames_rotating")
        return ms.value

    def time_each_frame(self, iters):
        """GPU milliseconds of each of `iters` frames (hipEvent pair per frame)."""
        ms = (C.c_float * iters)()
        _check(lib().rf_graph_time_each_frame(self._h, iters, ms), "rf_graph_time_each_frame")
        return list(ms)

    def time_launch(self, launch, iters):
        ms = C.c_float()
        _check(lib().rf_graph_time_launch(self._h, launch, iters, C.byref(ms)), "rf_graph_time_launch")
        return ms.value

    @property
    def note(self):
        """what rf_graph_create had to say about this graph ("" = nothing): catalogue-only fusion, a user stage that spills"""
        return _s(lib().rf_graph_note(self._h))

    def time_launches(self, iters):
        """[(label, average ms)] of every launch over `iters` frames, hipEvent pairs on the launch's stream."""
        labels = self.plan.launches()
        ms = (C.c_float * max(len(labels), 1))()
        _check(lib().rf_graph_time_launches(self._h, iters, ms, len(labels)), "rf_graph_time_launches")
        return list(zip(labels, list(ms[:len(labels)])))


@dataclass
class RenderInfo:
    """src/render.rs:37-48"""
    width: int
    height: int
    num_frames: int = 1
    config_path: Optional[str] = None
    shader_path: str = "shaders"          # where {type}.stage.hip is looked for when a type is not built in (config.rs:59-75)
    format: int = _lib.RF_FORMAT_RGBA32F
    swapchain: bool = False                # no display on an MI355X box
    has_input_image: bool = True
    shader_file_path: Optional[str] = None
    flags: int = 0
    device: int = 0


class Render:
    """Render (src/render.rs:50-58,:537-588): owns the config, the graph and the RGBA8
    staging buffer, and exposes the per-frame calls of the reference's render_fn
    (src/main.rs:134-182)."""

    DEFAULT_CONFIG = "input -> passthrough -> output"      # render.rs:115

    def __init__(self, info: RenderInfo, ctx: Optional[Context] = None):
        if info.swapchain:
            raise ValueError("swapchain presentation is out of scope (no display on an MI355X box)")
        self.info = info
        self.ctx = ctx or Context(info.device)
        self.frame_index = 0
        self.staging = np.zeros((info.height, info.width, 4), np.uint8)    # render.rs:552-555
        self._config_mtime = None
        self._stage_mtimes = {}
        self.graph = None
        if info.shader_path:
            set_shader_path(info.shader_path)
        self._create()
        if self.graph is None:
            raise ValueError("Unable to create config")                    # render.rs:543

    # create_config, render.rs:100-119
    def _load_config(self):
        info = self.info
        if info.config_path:
            try:
                with open(info.config_path) as fh:
                    text = fh.read()
            except OSError:
                return None
            self._config_mtime = os.path.getmtime(info.config_path)
            return Config(text, info.has_input_image)
        if info.shader_file_path:
            stem = os.path.splitext(os.path.basename(info.shader_file_path))[0]   # config.rs:79
            return Config(single=stem, expects_input=info.has_input_image)
        return Config(self.DEFAULT_CONFIG, info.has_input_image)

    def _create(self):
        """create_graph render.rs:80-98; on failure the previous graph keeps running (render.rs:121-136)."""
        try:
            cfg = self._load_config()
            if cfg is None:
                return False
            g = Graph(self.ctx, cfg, self.info.width, self.info.height, self.info.format,
                      self.info.num_frames, self.info.flags | _lib.RF_GRAPH_TIMERS)
        except RfError as e:
            if e.status in (_lib.RF_ERR_CONFIG, _lib.RF_ERR_GRAPH):
                return False
            raise
        if self.graph is not None:
            self.graph.close()
        self.graph = g
        # the stage files this graph was built from (reload_changed_pipelines, render.rs:225-249)
        types = {v["type"] for v in cfg.nodes().values()}
        self._stage_mtimes = {t: lib().rf_user_stage_mtime(t.encode()) for t in types if lib().rf_user_stage_mtime(t.encode()) >= 0}
        self._first_run = [True] * self.info.num_frames
        self._input_loaded = False
        return True

    def staging_buffer(self):                 # staging_buffer_ptr, render.rs:60
        return self.staging

    def wait_for_frame_fence(self):           # render.rs:328-337
        self.graph.wait(self.frame_index)

    def trigger_reloads(self):                # render.rs:497-519: the config file, and the stage files of user types (:225-249)
        changed = False
        p = self.info.config_path
        if p:
            try:
                m = os.path.getmtime(p)
            except OSError:
                m = 0
            if m != self._config_mtime:
                self._config_mtime = m
                changed = True
        for t, was in list(self._stage_mtimes.items()):
            try:
                now = os.stat(os.path.join(shader_path(), t + ".stage.hip")).st_mtime_ns
            except OSError:
                now = -1
            if now != was:
                self._stage_mtimes[t] = now              # a file that fails to build is not retried until it changes again
                changed = True
        return self._create() if changed else False

    def update_ubos(self, time_s):            # render.rs:212-223
        self.graph.set_time(time_s)

    def last_frame_gpu_times(self):           # render.rs:521-523
        return self.graph.times_string(self.frame_index)

    def record_initial_image_load(self):      # render.rs:264-313
        # rf_graph_upload_srgb8 writes the input image of EVERY frame slot, so it runs once per
        # graph (reforge_main.cpp's load_input does the same); the reference uploads per slot
        # because each of its slots owns a private input image (main.rs:164-170)
        self.graph.upload_srgb8(self.staging)
        self._input_loaded = True

    def record(self):                         # render.rs:359-404
        if self._first_run[self.frame_index] and self.info.has_input_image and not self._input_loaded:
            self.record_initial_image_load()
        self._first_run[self.frame_index] = False

    def submit(self):                         # render.rs:441-495
        self.graph.execute(self.frame_index)
        self._submitted = self.frame_index
        self.frame_index = (self.frame_index + 1) % self.info.num_frames

    def write_output_to_buffer(self):         # render.rs:406-433
        slot = getattr(self, "_submitted", 0)
        self.graph.wait(slot)
        self.staging[...] = self.graph.download_srgb8(slot)

    def render_frame(self, time_s=0.0):
        """One iteration of main.rs:134-182 in headless mode."""
        self.wait_for_frame_fence()
        self.trigger_reloads()
        self.update_ubos(time_s)
        self.record()
        self.submit()
        self.write_output_to_buffer()
        return self.staging
