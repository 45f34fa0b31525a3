"""ctypes binding of librfhip.so -- the C ABI declared in include/rfhip.h.

There is no fallback of any kind: if the shared library is missing this module
raises at import of the symbol table, and every device entry point fails with
RF_ERR_NO_DEVICE when no gfx950 GPU is usable.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "librfhip.so")

RF_OK = 0
RF_ERR_INVALID = 1
RF_ERR_CONFIG = 2
RF_ERR_GRAPH = 3
RF_ERR_NO_DEVICE = 4
RF_ERR_DEVICE = 5
RF_ERR_UNSUPPORTED = 6
RF_WARN_UNKNOWN_PARAM = 16

RF_FORMAT_RGBA8 = 0
RF_FORMAT_RGBA32F = 1

RF_PARAM_F32, RF_PARAM_I32, RF_PARAM_BOOL = 0, 1, 2

RF_GRAPH_TIMERS = 0x1
RF_GRAPH_NO_FUSION = 0x2
RF_GRAPH_HIPGRAPH = 0x4
RF_GRAPH_NO_HALO_XCHG = 0x8
RF_GRAPH_NO_JIT = 0x10
RF_GRAPH_GLSL_NODES = 0x20

RF_EXEC_SYNC_LAUNCHES = 0x1
RF_EXEC_CONCURRENT_LAYERS = 0x2
RF_EXEC_FORCE_SPLIT = 0x4
RF_EXEC_NO_ALTERNATE = 0x8
RF_EXEC_ALTERNATE = 0x20
RF_EXEC_GLSL_NO_WINDOW = 0x40

RF_CONV_AUTO, RF_CONV_TILE, RF_CONV_MFMA, RF_CONV_VALU = 0, 1, 2, 3


class GraphOptions(C.Structure):
    _fields_ = [("width", C.c_int), ("height", C.c_int), ("format", C.c_int),
                ("num_frames", C.c_int), ("flags", C.c_uint32),
                ("rows_per_chunk", C.c_int), ("conv_path", C.c_int), ("exec_flags", C.c_uint32),
                ("texels_per_lane", C.c_int)]


_vp, _cp, _i, _sz, _u32, _f = C.c_void_p, C.c_char_p, C.c_int, C.c_size_t, C.c_uint32, C.c_float
_pi, _pf, _pvp = C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_void_p)

# name -> (restype, argtypes): every symbol include/rfhip.h declares
SIGNATURES = {
    "rf_last_error": (_cp, []),
    "rf_abi_version": (_i, []),
    "rf_config_parse": (_i, [_cp, _i, _pvp]),
    "rf_config_single": (_i, [_cp, _i, _pvp]),
    "rf_config_syntax": (_i, [_cp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rf_config_destroy": (None, [_vp]),
    "rf_config_num_nodes": (_i, [_vp]),
    "rf_config_node_name": (_cp, [_vp, _i]),
    "rf_config_node_type": (_cp, [_vp, _i]),
    "rf_config_node_num_inputs": (_i, [_vp, _i]),
    "rf_config_node_num_outputs": (_i, [_vp, _i]),
    "rf_config_node_input_resource": (_cp, [_vp, _i, _i]),
    "rf_config_node_input_descriptor": (_cp, [_vp, _i, _i]),
    "rf_config_node_output_resource": (_cp, [_vp, _i, _i]),
    "rf_config_node_output_descriptor": (_cp, [_vp, _i, _i]),
    "rf_config_node_num_params": (_i, [_vp, _i]),
    "rf_config_node_param_key": (_cp, [_vp, _i, _i]),
    "rf_config_node_param_value": (_cp, [_vp, _i, _i]),
    "rf_plan_create": (_i, [_vp, _u32, _pvp]),
    "rf_plan_destroy": (None, [_vp]),
    "rf_plan_num_layers": (_i, [_vp]),
    "rf_plan_layer_size": (_i, [_vp, _i]),
    "rf_plan_layer_node": (_cp, [_vp, _i, _i]),
    "rf_plan_num_aliases": (_i, [_vp]),
    "rf_plan_alias_from": (_cp, [_vp, _i]),
    "rf_plan_alias_to": (_cp, [_vp, _i]),
    "rf_plan_num_images": (_i, [_vp]),
    "rf_plan_image_name": (_cp, [_vp, _i]),
    "rf_plan_resolve": (_cp, [_vp, _cp]),
    "rf_plan_num_buffers": (_i, [_vp]),
    "rf_plan_buffer_name": (_cp, [_vp, _i]),
    "rf_plan_buffer_bytes": (_sz, [_vp, _i]),
    "rf_plan_resolve_buffer": (_cp, [_vp, _cp]),
    "rf_plan_num_launches": (_i, [_vp]),
    "rf_plan_launch_label": (_cp, [_vp, _i]),
    "rf_plan_launch_layer": (_i, [_vp, _i]),
    "rf_plan_launch_num_members": (_i, [_vp, _i]),
    "rf_plan_launch_member": (_cp, [_vp, _i, _i]),
    "rf_plan_launch_member_slot": (_i, [_vp, _i, _i]),
    "rf_plan_launch_num_inputs": (_i, [_vp, _i]),
    "rf_plan_launch_input": (_cp, [_vp, _i, _i]),
    "rf_plan_launch_output": (_cp, [_vp, _i]),
    "rf_plan_launch_num_outputs": (_i, [_vp, _i]),
    "rf_plan_launch_output_at": (_cp, [_vp, _i, _i]),
    "rf_plan_launch_radius": (_i, [_vp, _i]),
    "rf_plan_launch_serial": (_i, [_vp, _i]),
    "rf_plan_signature": (C.c_uint64, [_vp]),
    "rf_set_shader_path": (_i, [_cp]),
    "rf_shader_path": (_cp, []),
    "rf_user_stage_mtime": (C.c_longlong, [_cp]),
    "rf_glsl_translate": (_i, [_cp, _cp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rf_glsl_reflect": (_i, [_cp, _cp, C.c_char_p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "rf_set_type_lookup": (_i, [C.c_int]),
    "rf_type_lookup": (C.c_int, []),
    "rf_jit_available": (_i, []),
    "rf_jit_compile_count": (_i, []),
    "rf_jit_library": (_cp, []),
    "rf_plan_launch_needs_jit": (_i, [_vp, _i]),
    "rf_plan_jit_compile": (_i, [_vp, _i, C.POINTER(C.c_size_t)]),
    "rf_plan_jit_compile_texels": (_i, [_vp, _i, _i, C.POINTER(C.c_size_t)]),
    "rf_plan_halo_schedule": (_i, [_vp, _i, _pi, _pi, _i, _pi, _pi]),
    "rf_registry_num_types": (_i, []),
    "rf_registry_type_name": (_cp, [_i]),
    "rf_registry_buffer_binding": (_i, [_cp, _cp]),
    "rf_registry_binding": (_i, [_cp, _cp]),
    "rf_registry_radius": (_i, [_cp]),
    "rf_strip_rows": (_i, [_i, _i, _i, _pi, _pi]),
    "rf_ctx_create": (_i, [_i, _pvp]),
    "rf_comm_unique_id": (_i, [_vp]),
    "rf_comm_library": (_cp, []),
    "rf_ctx_create_dist": (_i, [_i, _i, _i, _vp, _pvp]),
    "rf_ctx_destroy": (None, [_vp]),
    "rf_ctx_synchronize": (_i, [_vp]),
    "rf_ctx_rank": (_i, [_vp]),
    "rf_ctx_world": (_i, [_vp]),
    "rf_ctx_device_arch": (_cp, [_vp]),
    "rf_graph_create": (_i, [_vp, _vp, C.POINTER(GraphOptions), _pvp]),
    "rf_graph_destroy": (None, [_vp]),
    "rf_graph_plan": (_vp, [_vp]),
    "rf_graph_strip": (_i, [_vp, _pi, _pi]),
    "rf_graph_set_param": (_i, [_vp, _cp, _cp, _i, _vp]),
    "rf_graph_set_weights": (_i, [_vp, _cp, _pf, _i]),
    "rf_graph_set_time": (_i, [_vp, _f]),
    "rf_graph_upload_srgb8": (_i, [_vp, _vp, _sz]),
    "rf_graph_upload_raw": (_i, [_vp, _vp, _sz]),
    "rf_graph_fill_synthetic": (_i, [_vp, _u32]),
    "rf_graph_fill_structured": (_i, [_vp]),
    "rf_graph_execute": (_i, [_vp, _i]),
    "rf_graph_wait": (_i, [_vp, _i]),
    "rf_graph_download_srgb8": (_i, [_vp, _i, _vp, _sz]),
    "rf_graph_download_raw": (_i, [_vp, _i, _vp, _sz]),
    "rf_graph_download_image": (_i, [_vp, _i, _cp, _vp, _sz]),
    "rf_graph_download_rows": (_i, [_vp, _i, _i, _i, _vp, _sz]),
    "rf_graph_node_times": (_i, [_vp, _i, C.POINTER(_cp), _pf, _pi]),
    "rf_graph_times_string": (_i, [_vp, _i, _cp, _sz]),
    "rf_graph_time_frames": (_i, [_vp, _i, _pf]),
    "rf_graph_time_frames_rotating": (_i, [_vp, _i, _pf]),
    "rf_graph_time_each_frame": (_i, [_vp, _i, _pf]),
    "rf_graph_time_launch": (_i, [_vp, _i, _i, _pf]),
    "rf_graph_time_launches": (_i, [_vp, _i, _pf, _i]),
    "rf_graph_note": (C.c_char_p, [_vp]),
    "rf_comm_selftest": (_i, [_i, _sz]),
    "rf_ctx_copy_bandwidth": (_i, [_vp, _sz, _i, _pf]),
}

_lib = None


def lib():
    """The loaded library with argtypes set.  Raises if librfhip.so is not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise ImportError(
                "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C reforge_amd/csrc`.  reforge_amd has no CPU fallback." % SO_PATH)
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)      # AttributeError if the ABI and the library disagree
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib
