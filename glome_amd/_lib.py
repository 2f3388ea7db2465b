"""ctypes binding of libglome_hip.so (include/glome_hip.h).  There is no fallback: if the library has not been
built (python glome_amd/build.py, or __graft_entry__.build()) importing the product API raises."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# GLOME_DEBUG_LIB: load another build of the same library (kernel experiments, tools/variant_bench.py); never a fallback
LIB_PATH = os.environ.get("GLOME_DEBUG_LIB") or os.path.join(HERE, "libglome_hip.so")

c_dp = C.POINTER(C.c_double)
c_fp = C.POINTER(C.c_float)
c_ip = C.POINTER(C.c_int32)
c_up = C.POINTER(C.c_uint32)
c_bp = C.POINTER(C.c_uint8)


class Camera(C.Structure):  # glome_camera
    _fields_ = [("pos", C.c_float * 3), ("fwd", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3)]


class Light(C.Structure):  # glome_light
    _fields_ = [("pos", C.c_float * 3), ("color", C.c_float * 3), ("rad", C.c_float), ("shadow", C.c_int32)]


OK, E_INVALID, E_SCENE, E_NO_DEVICE, E_HIP, E_LIMIT = 0, -1, -2, -3, -4, -5  # glome_status


class RenderParams(C.Structure):  # glome_render_params
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("mode", C.c_int32), ("blocksize", C.c_int32),
                ("maxdepth", C.c_int32), ("fog", C.c_int32), ("thresholds", C.c_float * 4), ("tile_first", C.c_int32),
                ("tile_stride", C.c_int32), ("faithful", C.c_int32), ("count_work", C.c_int32), ("rank0_share_pct", C.c_int32)]


class Stats(C.Structure):  # glome_stats
    _fields_ = [("rays_primary", C.c_uint64), ("rays_shadow", C.c_uint64), ("rays_secondary", C.c_uint64),
                ("bih_nodes", C.c_uint64), ("mesh_nodes", C.c_uint64), ("prim_tests", C.c_uint64),
                ("kernel_ms", C.c_float), ("n_tiles", C.c_int32), ("n_pixels", C.c_int32)]


class SceneInfo(C.Structure):  # glome_scene_info
    _fields_ = [("tier", C.c_int32), ("nesting_depth", C.c_int32), ("n_records", C.c_int64), ("n_bih_nodes", C.c_int64),
                ("n_mesh_nodes", C.c_int64), ("n_triangles", C.c_int64), ("n_spheres", C.c_int64),
                ("n_other_prims", C.c_int64), ("n_xfms", C.c_int64), ("n_materials", C.c_int64),
                ("max_bih_depth", C.c_int32), ("max_mesh_depth", C.c_int32), ("device_bytes", C.c_int64)]


# every symbol include/glome_hip.h declares: (name, restype, argtypes)
vp = C.c_void_p
SYMBOLS = [
    ("glome_ctx_create", vp, [C.c_int]),
    ("glome_ctx_destroy", None, [vp]),
    ("glome_last_error", C.c_char_p, [vp]),
    ("glome_global_error", C.c_char_p, []),
    ("glome_ctx_stream", vp, [vp]),
    ("glome_ctx_synchronize", C.c_int, [vp]),
    ("glome_ctx_use_stream", C.c_int, [vp, vp]),
    ("glome_ctx_use_slot", C.c_int, [vp, vp, C.c_int]),
    ("glome_ctx_timing_begin", C.c_int, [vp, C.c_int]),
    ("glome_ctx_timing_begin_sampled", C.c_int, [vp, C.c_int, C.c_int]),
    ("glome_ctx_timing_end", C.c_int, [vp, c_fp, C.c_int]),
    ("glome_ctx_set_grid_per_cu", C.c_int, [vp, C.c_int]),
    ("glome_ipc_alloc", C.c_int, [vp, C.c_size_t, C.POINTER(vp), C.c_char_p]),
    ("glome_ipc_open", C.c_int, [vp, C.c_char_p, C.POINTER(vp)]),
    ("glome_ipc_close", C.c_int, [vp, vp, C.c_int]),
    ("glome_multi_create", vp, [C.POINTER(vp), C.c_int, vp, C.c_int]),
    ("glome_multi_destroy", None, [vp]),
    ("glome_multi_render", C.c_int, [vp, vp, C.c_int, vp, C.c_int, vp]),
    ("glome_multi_synchronize", C.c_int, [vp]),
    ("glome_multi_transport", C.c_char_p, [vp]),
    ("glome_multi_last_error", C.c_char_p, [vp]),
    ("glome_render_multi", C.c_int, [C.POINTER(vp), C.c_int, vp, vp, C.c_int, vp, vp]),
    ("glome_ctx_device_info", C.c_int, [vp, C.c_char_p, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("glome_ctx_debug_words", C.c_int, [vp, C.POINTER(C.c_uint64)]),
    ("glome_ctx_debug_reset", C.c_int, [vp]),
    ("glome_xfm_translate", C.c_int, [c_dp, c_dp]),
    ("glome_xfm_scale", C.c_int, [c_dp, c_dp]),
    ("glome_xfm_rotate", C.c_int, [c_dp, C.c_double, c_dp]),
    ("glome_xfm_xyz_to_uvw", C.c_int, [c_dp, c_dp, c_dp, c_dp]),
    ("glome_xfm_compose", C.c_int, [c_dp, C.c_int, c_dp]),
    ("glome_sb_new", vp, []),
    ("glome_sb_free", None, [vp]),
    ("glome_sb_last_error", C.c_char_p, [vp]),
    ("glome_sb_sphere", C.c_int32, [vp, c_dp, C.c_double]),
    ("glome_sb_triangle", C.c_int32, [vp, c_dp]),
    ("glome_sb_trianglenorm", C.c_int32, [vp, c_dp, c_dp]),
    ("glome_sb_box", C.c_int32, [vp, c_dp, c_dp]),
    ("glome_sb_plane", C.c_int32, [vp, c_dp, c_dp]),
    ("glome_sb_plane_offset", C.c_int32, [vp, c_dp, C.c_double]),
    ("glome_sb_disc", C.c_int32, [vp, c_dp, c_dp, C.c_double]),
    ("glome_sb_cylinder", C.c_int32, [vp, c_dp, c_dp, C.c_double]),
    ("glome_sb_cone", C.c_int32, [vp, c_dp, C.c_double, c_dp, C.c_double]),
    ("glome_sb_group", C.c_int32, [vp, c_ip, C.c_int]),
    ("glome_sb_transform", C.c_int32, [vp, C.c_int32, c_dp, C.c_int]),
    ("glome_tex_words", C.c_int, []),
    ("glome_sb_difference", C.c_int32, [vp, C.c_int32, C.c_int32]),
    ("glome_sb_difference_retexture", C.c_int32, [vp, C.c_int32, C.c_int32]),
    ("glome_sb_intersection", C.c_int32, [vp, c_ip, C.c_int]),
    ("glome_sb_bih", C.c_int32, [vp, c_ip, C.c_int]),
    ("glome_sb_mesh", C.c_int32, [vp, c_dp, C.c_int, c_dp, C.c_int, c_ip, C.c_int, c_ip, C.c_int]),
    ("glome_sb_tex", C.c_int32, [vp, C.c_int32, C.c_int32]),
    ("glome_sb_tag", C.c_int32, [vp, C.c_int32]),
    ("glome_sb_noshadow", C.c_int32, [vp, C.c_int32]),
    ("glome_sb_onlyshadow", C.c_int32, [vp, C.c_int32]),
    ("glome_sb_bound_object", C.c_int32, [vp, C.c_int32, C.c_int32]),
    ("glome_sb_innerbound", C.c_int32, [vp, C.c_int32, C.c_int32]),
    ("glome_sb_flatten_transform", C.c_int32, [vp, C.c_int32]),
    ("glome_sb_tolist", C.c_int32, [vp, C.c_int32]),
    ("glome_sb_list_items", C.c_int32, [vp, C.c_int32, c_ip, C.c_int32]),
    ("glome_sb_material_surface", C.c_int32, [vp, c_dp, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double]),
    ("glome_sb_material_reflect", C.c_int32, [vp, C.c_double]),
    ("glome_sb_material_refract", C.c_int32, [vp, C.c_double, C.c_double, C.c_double]),
    ("glome_sb_material_layers", C.c_int32, [vp, c_ip, C.c_int]),
    ("glome_sb_material_blend", C.c_int32, [vp, C.c_int32, C.c_int32, C.c_double]),
    ("glome_sb_material_blend_fn", C.c_int32, [vp, C.c_int32, C.c_int32, C.c_int32, c_dp]),
    ("glome_sb_material_warp", C.c_int32, [vp, C.c_int32, C.c_int32, vp, C.c_int, c_dp]),
    ("glome_sb_load_nff", C.c_int32, [vp, C.c_char_p, c_dp, c_dp, C.c_int32, c_ip, c_dp]),
    ("glome_sb_bih_dev", C.c_int32, [vp, vp, c_ip, C.c_int32, C.POINTER(C.c_float)]),
    ("glome_sb_mesh_dev", C.c_int32, [vp, vp, c_dp, C.c_int, c_dp, C.c_int, c_ip, C.c_int, c_ip, C.c_int, C.POINTER(C.c_float)]),
    ("glome_sb_show", C.c_long, [vp, C.c_int32, C.c_char_p, C.c_long]),
    ("glome_sb_show_tex_materials", C.c_long, [vp, C.c_int32, c_ip, C.c_long]),
    ("glome_sb_load_show", C.c_int32, [vp, C.c_char_p, c_ip, C.c_int32, C.c_int32, c_ip]),
    ("glome_sb_primcount", C.c_int, [vp, C.c_int32, C.POINTER(C.c_long)]),
    ("glome_sb_bound", C.c_int, [vp, C.c_int32, c_dp]),
    ("glome_sb_bih_dump", C.c_long, [vp, C.c_int32, C.c_long, c_dp, c_dp, C.POINTER(C.c_int), C.POINTER(C.c_int), c_ip, C.c_long]),
    ("glome_scene_commit", vp, [vp, vp, C.c_int32]),
    ("glome_scene_release", None, [vp]),
    ("glome_scene_get_info", C.c_int, [vp, C.POINTER(SceneInfo)]),
    ("glome_rayint_batch", C.c_int, [vp, C.c_size_t] + [c_fp] * 7 + [c_fp, c_ip, c_fp, c_fp, c_fp, c_ip]),
    ("glome_shadow_batch", C.c_int, [vp, C.c_size_t] + [c_fp] * 7 + [c_bp]),
    ("glome_inside_batch", C.c_int, [vp, C.c_size_t, c_fp, c_fp, c_fp, c_bp]),
    ("glome_rayint_batch_dev", C.c_int, [vp, C.c_size_t] + [vp] * 13),
    ("glome_shadow_batch_dev", C.c_int, [vp, C.c_size_t] + [vp] * 8),
    ("glome_camera_lookat", C.c_int, [c_dp, c_dp, c_dp, C.c_double, C.POINTER(Camera)]),
    ("glome_render_params_default", None, [C.POINTER(RenderParams)]),
    ("glome_render", C.c_int, [vp, C.POINTER(Camera), C.POINTER(Light), C.c_int, C.POINTER(RenderParams), c_fp, c_up, C.POINTER(Stats)]),
    ("glome_render_dev", C.c_int, [vp, C.POINTER(Camera), C.POINTER(Light), C.c_int, C.POINTER(RenderParams), vp, vp, C.POINTER(Stats)]),
    ("glome_render_tiles_dev", C.c_int, [vp, C.POINTER(Camera), C.POINTER(Light), C.c_int, C.POINTER(RenderParams), vp, C.POINTER(Stats)]),
    ("glome_tiles_payload_floats", C.c_int64, [C.POINTER(RenderParams), C.c_int, C.c_int]),
    ("glome_tiles_layout", C.c_int, [C.POINTER(RenderParams), C.c_int, C.c_int, c_ip, C.c_int]),
    ("glome_tiles_pack_dev", C.c_int, [vp, C.POINTER(RenderParams), vp, vp]),
    ("glome_tiles_blit_dev", C.c_int, [vp, C.POINTER(RenderParams), C.c_int, C.c_int, vp, vp, vp]),
    ("glome_tiles_blit_all_dev", C.c_int, [vp, C.POINTER(RenderParams), C.c_int, vp, C.c_int64, vp, vp]),
    ("glome_render_tiles_packed_dev", C.c_int, [vp, C.POINTER(Camera), C.POINTER(Light), C.c_int, C.POINTER(RenderParams), vp, C.POINTER(Stats)]),
    ("glome_tiles_blit_all_packed_dev", C.c_int, [vp, C.POINTER(RenderParams), C.c_int, vp, C.c_int64, vp]),
    ("glome_tiles_blit_all_packed_batch_dev", C.c_int, [vp, C.POINTER(RenderParams), C.c_int, vp, C.c_int64, C.c_int, C.c_int64, vp, C.c_int64]),
    ("glome_render_tiles_packed_batch_dev", C.c_int, [vp, C.POINTER(Camera), C.c_int, C.POINTER(Light), C.c_int, C.POINTER(RenderParams), vp, C.c_int64, C.POINTER(Stats)]),
    ("glome_render_packed_batch_dev", C.c_int, [vp, C.POINTER(Camera), C.c_int, C.POINTER(Light), C.c_int, C.POINTER(RenderParams), vp, C.c_int64, C.POINTER(Stats)]),
]

_lib = None


def load():
    """Load libglome_hip.so and bind every declared symbol.  Raises if the extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension has not been built (run `python glome_amd/build.py`). "
            "glome_amd has no CPU fallback.")
    # torch bundles its own HIP runtime under the same soname (libamdhip64.so.7).  Two HIP runtimes cannot share
    # one process, so when torch is installed let it load first; this library then binds to the same runtime and
    # torch tensors / torch.distributed (RCCL) can be used beside it.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        if os.environ.get("GLOME_DEBUG_LIB") and not hasattr(lib, name):
            continue  # (an older build of the library loaded for an A/B measurement may lack entries that came later)
        f = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def dvec(seq):
    a = np.ascontiguousarray(np.asarray(seq, dtype=np.float64).ravel())
    return a, a.ctypes.data_as(c_dp)


def ivec(seq):
    a = np.ascontiguousarray(np.asarray(seq, dtype=np.int32).ravel())
    return a, a.ctypes.data_as(c_ip)
