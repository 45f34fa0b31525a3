"""reforge_amd -- MI355X-native executor for reforge's image-filter render graph.

The product is the C-ABI library `librfhip.so` (include/rfhip.h): hand-written
gfx950 HIP kernels + a stream/event graph executor + RCCL row-strip halo exchange.
This package is the thin Python host over it (ctypes), used by the tests, bench.py
and as a scripting front end; the C++ CLI host is `reforge_amd/reforge`
(csrc/host/reforge_main.cpp).

Names mirror the reference: Config (src/config/config.rs), the plan of
PipelineGraph (src/vulkan/pipeline_graph.rs), Render / RenderInfo
(src/render.rs:37-58,:537-588).
"""
from ._lib import (RF_FORMAT_RGBA8, RF_FORMAT_RGBA32F, RF_GRAPH_TIMERS, RF_GRAPH_NO_FUSION,
                   RF_GRAPH_HIPGRAPH, RF_GRAPH_NO_HALO_XCHG, RF_GRAPH_NO_JIT, RF_GRAPH_GLSL_NODES, RF_EXEC_SYNC_LAUNCHES, RF_EXEC_CONCURRENT_LAYERS,
                   RF_EXEC_FORCE_SPLIT, RF_EXEC_NO_ALTERNATE, RF_EXEC_ALTERNATE, RF_EXEC_GLSL_NO_WINDOW, RF_CONV_AUTO, RF_CONV_TILE,
                   RF_CONV_MFMA, RF_CONV_VALU, SO_PATH, lib)
from .host import (RfError, Config, Plan, Context, Graph, Render, RenderInfo, get_dim, comm_selftest,
                   registry_types, registry_binding, strip_rows, set_shader_path, shader_path, config_syntax, FILE_INPUT, FINAL_OUTPUT,
                   glsl_translate, glsl_reflect, set_type_lookup)

__all__ = [
    "RF_FORMAT_RGBA8", "RF_FORMAT_RGBA32F", "RF_GRAPH_TIMERS", "RF_GRAPH_NO_FUSION",
    "RF_GRAPH_HIPGRAPH", "RF_GRAPH_NO_HALO_XCHG", "RF_GRAPH_NO_JIT", "RF_GRAPH_GLSL_NODES", "RF_EXEC_SYNC_LAUNCHES", "RF_EXEC_CONCURRENT_LAYERS",
    "RF_EXEC_FORCE_SPLIT", "RF_EXEC_NO_ALTERNATE", "RF_EXEC_ALTERNATE", "RF_EXEC_GLSL_NO_WINDOW", "RF_CONV_AUTO", "RF_CONV_TILE", "RF_CONV_MFMA",
    "RF_CONV_VALU", "SO_PATH", "lib",
    "RfError", "Config", "Plan", "Context", "Graph", "Render", "RenderInfo", "get_dim",
    "registry_types", "registry_binding", "strip_rows", "set_shader_path", "shader_path", "config_syntax", "FILE_INPUT", "FINAL_OUTPUT",
    "glsl_translate", "glsl_reflect", "set_type_lookup",
]
