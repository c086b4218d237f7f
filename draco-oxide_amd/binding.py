"""ctypes binding of libdraco_mi.so.  No arithmetic lives here; see include/draco_mi.h for the ABI.

Reference interface mirrored (paths relative to draco-oxide/src/):
  encode(mesh, writer, cfg)      encode/mod.rs:59      (Mesh consumed, bytes appended to the writer)
  Config.default()               encode/mod.rs:32-42   (only the default configuration is public)
  Mesh / Attribute / MeshBuilder core/mesh/mod.rs:13-23, core/attribute/mod.rs:26-49, core/mesh/builder.rs:14-90
"""
import ctypes as C
import os
import sys
import subprocess

import numpy as np

ATT_POSITION, ATT_NORMAL, ATT_COLOR, ATT_TEXCOORD, ATT_CUSTOM = 0, 1, 2, 3, 4
DOMAIN_POSITION, DOMAIN_CORNER = 0, 1
U32, I32, F32 = 5, 6, 9
FLAG_TIMINGS = 1
POS_SCHEME_DELTA = 0xD0
NONE = 0xFFFFFFFF

_PKG = os.path.dirname(os.path.abspath(__file__))
_LIB = os.environ.get("DMI_LIBRARY") or os.path.join(_PKG, "libdraco_mi.so")   # DMI_LIBRARY: A/B runs against another in-tree build
_lib = None


class DracoMiError(RuntimeError):
    def __init__(self, status, what, detail):
        super().__init__(f"libdraco_mi status {status} ({what}): {detail}")
        self.status = status


class _Attribute(C.Structure):
    _fields_ = [("values", C.c_void_p), ("num_unique", C.c_uint32), ("component_type", C.c_uint8), ("num_components", C.c_uint8),
                ("att_type", C.c_uint8), ("domain", C.c_uint8), ("unique_id", C.c_uint32), ("parent_index", C.c_int32),
                ("point_to_value", C.c_void_p), ("num_points", C.c_uint32)]


class _CornerTable(C.Structure):
    _fields_ = [("num_faces", C.c_uint32), ("num_vertices", C.c_uint32), ("corner_to_point", C.c_void_p), ("corner_to_vertex", C.c_void_p),
                ("opposite", C.c_void_p), ("left_most_corner", C.c_void_p), ("sequence", C.c_void_p), ("sequence_len", C.c_uint32)]


class _Debug(C.Structure):
    """dmi_debug (include/draco_mi.h): the typed path-selecting / tuning switches.  The LIBRARY never reads the environment for them (round 6); this
    binding fills the struct from os.environ on every call — `DMI_NO_FUSED=1 pytest …` and monkeypatch.setenv keep working for tests and bench."""
    _fields_ = [("flags", C.c_uint64), ("host_threads", C.c_uint32), ("tile_sort", C.c_int32), ("tile_sort_min", C.c_uint32), ("tile_sort_local", C.c_uint32),
                ("seq_big_entries", C.c_uint32), ("relabel", C.c_uint8), ("chains", C.c_uint8), ("prepare_threads", C.c_uint8), ("pad1", C.c_uint8),
                ("fused_grid", C.c_uint32), ("fused_lds", C.c_uint32), ("chain_grid", C.c_uint32), ("batch_threads", C.c_uint32), ("split", C.c_uint32),
                ("shadow_min_faces", C.c_uint32), ("prep_group_faces", C.c_uint64), ("batch_min_faces", C.c_uint64), ("stage_primitives", C.c_uint32), ("stage_ramp", C.c_uint32)]


class _Config(C.Structure):
    _fields_ = [("pos_bits", C.c_uint8), ("uv_bits", C.c_uint8), ("generic_bits", C.c_uint8), ("pos_scheme", C.c_uint8),
                ("device", C.c_int32), ("stream", C.c_void_p), ("flags", C.c_uint32), ("debug", C.POINTER(_Debug))]


class _ProcessOptions(C.Structure):
    _fields_ = [("flags", C.c_uint32), ("host_cache_mb", C.c_uint32), ("device_cache_mb", C.c_uint32), ("decode_budget_mb", C.c_uint32)]


# environment variable → bit of dmi_debug.flags (set when the variable is present, whatever its value — what the library's getenv tests did)
_DEBUG_FLAGS = {"DMI_NO_FUSED": 0, "DMI_NO_PACKED": 1, "DMI_NO_SYM16": 2, "DMI_NO_EARLY": 3, "DMI_NO_PLAIN_ORDER": 4, "DMI_HOST_TABLES": 5, "DMI_HOST_CONNECTIVITY": 6,
                "DMI_HOST_ATT_TABLES": 7, "DMI_HOST_BUILD": 8, "DMI_NO_IN_PLACE": 9, "DMI_NO_POOL": 10, "DMI_POISON": 11, "DMI_ZERO_CHUNKS": 12, "DMI_NO_QUAD": 13,
                "DMI_TEST_QUAD": 14, "DMI_NO_CLOSED": 15, "DMI_NO_SHADOW": 16, "DMI_NO_SEQ_SHADOW": 17, "DMI_NO_SEAM_MASKS": 18, "DMI_NO_DEFER_SEAMS": 19,
                "DMI_NO_BATCHED_PHASES": 20, "DMI_FUSED_WINDOWS": 21, "DMI_CHAIN_DENSE": 22, "DMI_SERIAL_TABLES": 23, "DMI_PARALLEL_TABLES": 24, "DMI_FILE_ORDER": 25,
                "DMI_TRACE": 26, "DMI_TRACE_STAGES": 27, "DMI_TRACE_TABLES": 28, "DMI_BUILD_TRACE": 29, "DMI_NO_SEQ_STREAM": 30, "DMI_SMALL_HEAD": 31, "DMI_SPIN_WAITS": 32, "DMI_NO_STREAM_COPY": 33}
_DEBUG_INTS = {"DMI_HOST_THREADS": "host_threads", "DMI_TILE_SORT_MIN": "tile_sort_min", "DMI_TILE_SORT_LOCAL": "tile_sort_local", "DMI_SEQ_BIG_ENTRIES": "seq_big_entries",
               "DMI_FUSED_GRID": "fused_grid", "DMI_FUSED_LDS": "fused_lds", "DMI_CHAIN_GRID": "chain_grid", "DMI_BATCH_THREADS": "batch_threads",
               "DMI_SHADOW_MIN_FACES": "shadow_min_faces", "DMI_PREP_GROUP_FACES": "prep_group_faces", "DMI_BATCH_MIN_FACES": "batch_min_faces",
               "DMI_STAGE_PRIMITIVES": "stage_primitives", "DMI_STAGE_RAMP": "stage_ramp", "DMI_PREPARE_THREADS": "prepare_threads"}


def debug_from_env(env=None):
    """The dmi_debug struct the environment asks for (DMI_* variables: INTEGRATION.md §4)."""
    env = os.environ if env is None else env
    d = _Debug()
    flags = 0
    for name, bit in _DEBUG_FLAGS.items():
        if name in env:
            flags |= 1 << bit
    d.flags = flags
    for name, field in _DEBUG_INTS.items():
        v = env.get(name)
        if v:
            try:
                n = max(0, int(v))
            except ValueError:
                continue
            if n == 0 and name in ("DMI_SEQ_BIG_ENTRIES", "DMI_TILE_SORT_MIN", "DMI_SHADOW_MIN_FACES"):
                n = 1                                   # 0 in the struct means "default": a threshold of 0 is asked for as 1 (every real input is above it)
            setattr(d, field, n)
    if "DMI_SPLIT" in env:
        d.split = 1
    ts = env.get("DMI_TILE_SORT")
    if ts is not None:
        try:
            d.tile_sort = -1 if int(ts) <= 0 else int(ts)      # 0 = default in the struct, -1 = off
        except ValueError:
            pass
    d.relabel = {"device": 1, "host": 2}.get(env.get("DMI_RELABEL", ""), 0)
    d.chains = {"device": 1, "host": 2}.get(env.get("DMI_CHAINS", ""), 0)
    return d


def _sync_default_debug(L):
    """Entry points without a dmi_config (dmi_encode_connectivity, dmi_mesh_build, the host coders …) work under the process default: kept equal to the
    environment's, like every config this binding marshals."""
    d = debug_from_env()
    L.dmi_set_default_debug(C.byref(d))


def configure_process(huge_page_new=False, numa_pin=False, no_thp=False, host_cache_mb=0, device_cache_mb=0, decode_budget_mb=0):
    """dmi_configure_process: what the library does to its host process only when asked (round 6): large operator-new blocks of its own containers on huge
    pages, the calling thread restricted to the GPU's memory node during whole-mesh calls.  bench.py turns both on; the tests run with both off."""
    o = _ProcessOptions((1 if huge_page_new else 0) | (2 if numa_pin else 0) | (4 if no_thp else 0), int(host_cache_mb), int(device_cache_mb), int(decode_budget_mb))
    _check(load_library().dmi_configure_process(C.byref(o)))


class _Buffer(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", C.c_size_t), ("cap", C.c_size_t)]


class _Timings(C.Structure):
    _fields_ = [("quantize_ms", C.c_float), ("predict_ms", C.c_float), ("histogram_ms", C.c_float), ("table_ms", C.c_float),
                ("rans_ms", C.c_float), ("total_ms", C.c_float), ("predict_bytes", C.c_uint64), ("symbols", C.c_uint64), ("num_streams", C.c_uint32),
                ("host_chains", C.c_uint32), ("longest_stream_ms", C.c_float), ("readback_wait_ms", C.c_float),
                ("mesh_readback_ms", C.c_float), ("tables_ms", C.c_float), ("connectivity_ms", C.c_float), ("job_create_ms", C.c_float), ("call_ms", C.c_float), ("texcoord_fixups", C.c_uint32), ("job_create_device_ms", C.c_float), ("early_ms", C.c_float)]


class _Mesh(C.Structure):
    _fields_ = [("faces", C.c_void_p), ("num_faces", C.c_uint32), ("atts", C.POINTER(_Attribute)), ("num_atts", C.c_uint32)]


class _BatchItem(C.Structure):
    _fields_ = [("atts", C.POINTER(_Attribute)), ("tables", C.POINTER(_CornerTable)), ("n_atts", C.c_uint32), ("seeds", C.c_void_p), ("n_seeds", C.c_uint32)]


class _RawAttribute(C.Structure):
    _fields_ = [("data", C.c_void_p), ("num_points", C.c_uint32), ("component_type", C.c_uint8), ("num_components", C.c_uint8),
                ("att_type", C.c_uint8), ("domain", C.c_uint8), ("num_parents", C.c_uint32), ("parents", C.c_void_p)]


class _BuiltMesh(C.Structure):
    _fields_ = [("mesh", _Mesh), ("owner", C.c_void_p)]


class _RawAccessor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("count", C.c_uint32), ("byte_stride", C.c_uint32), ("component_type", C.c_uint8), ("num_components", C.c_uint8),
                ("att_type", C.c_uint8), ("domain", C.c_uint8), ("num_parents", C.c_uint32), ("parents", C.c_void_p)]


class _RawMesh(C.Structure):
    _fields_ = [("atts", C.POINTER(_RawAccessor)), ("n_atts", C.c_uint32), ("indices", C.c_void_p), ("index_type", C.c_uint8), ("num_faces", C.c_uint32)]


class _BuildTimings(C.Structure):
    _fields_ = [("pack_ms", C.c_float), ("kernels_ms", C.c_float), ("call_ms", C.c_float), ("device_meshes", C.c_uint32), ("host_meshes", C.c_uint32),
                ("bytes_up", C.c_uint64), ("bytes_down", C.c_uint64), ("in_place_meshes", C.c_uint32), ("pad", C.c_uint32)]


class _Span(C.Structure):
    _fields_ = [("data", C.c_void_p), ("bytes", C.c_size_t)]


class _GltfAsset(C.Structure):
    _fields_ = [("glb", C.c_void_p), ("glb_bytes", C.c_size_t), ("json", C.c_void_p), ("json_bytes", C.c_size_t), ("buffers", C.POINTER(_Span)), ("n_buffers", C.c_uint32)]


class _TranscodeStats(C.Structure):
    _fields_ = [("files", C.c_uint32), ("primitives", C.c_uint32), ("devices", C.c_uint32), ("buffers_in_place", C.c_uint32),
                ("primitives_device_built", C.c_uint32), ("primitives_host_built", C.c_uint32), ("primitives_in_place", C.c_uint32), ("pad", C.c_uint32),
                ("triangles_in", C.c_uint64), ("bytes_in", C.c_uint64), ("bytes_out", C.c_uint64),
                ("parse_ms", C.c_double), ("pushed_ms", C.c_double), ("finished_ms", C.c_double),
                ("build_ms", C.c_double), ("prepare_ms", C.c_double), ("encode_ms", C.c_double), ("assemble_ms", C.c_double), ("call_ms", C.c_double),
                ("parse_cpu_ms", C.c_double), ("parse_threads", C.c_uint32), ("stages", C.c_uint32)]


class _DecodedAttribute(C.Structure):
    _fields_ = [("att_type", C.c_uint8), ("component_type", C.c_uint8), ("num_components", C.c_uint8), ("domain", C.c_uint8),
                ("scheme", C.c_uint8), ("transform", C.c_uint8), ("portabilization", C.c_uint8), ("bits", C.c_uint8),
                ("unique_id", C.c_uint32), ("num_points", C.c_uint32), ("values", C.c_void_p)]


class _Decoded(C.Structure):
    _fields_ = [("num_attributes", C.c_uint32), ("attributes", C.POINTER(_DecodedAttribute)), ("owner", C.c_void_p)]


class _DecodeTimings(C.Structure):
    _fields_ = [(k, C.c_float) for k in ("connectivity_ms", "tables_ms", "sequence_ms", "entropy_ms", "inverse_ms", "device_ms", "attributes_ms", "call_ms")]


class _DecodedMesh(C.Structure):
    _fields_ = [("num_faces", C.c_uint32), ("num_points", C.c_uint32), ("faces", C.POINTER(C.c_uint32)), ("num_attributes", C.c_uint32),
                ("attributes", C.POINTER(_DecodedAttribute)), ("owner", C.c_void_p)]


class _Conn(C.Structure):
    _fields_ = [("num_tables", C.c_uint32), ("tables", C.POINTER(_CornerTable)), ("seeds", C.c_void_p), ("num_seeds", C.c_uint32), ("owner", C.c_void_p)]


_TRANSCODE_DONE = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, C.c_uint32)   # dmi_transcode_done_fn
EXPORTS = ["dmi_encode_attributes", "dmi_encode_attributes_batch", "dmi_jobs_encode", "dmi_job_create", "dmi_job_encode", "dmi_job_timings", "dmi_job_destroy", "dmi_encode_mesh",
           "dmi_mesh_prepare", "dmi_meshes_prepare", "dmi_shard_meshes", "dmi_meshes_prepare_devices", "dmi_jobs_encode_devices", "dmi_mesh_build", "dmi_built_mesh_free", "dmi_encode_connectivity", "dmi_conn_free", "dmi_host_rans_stream", "dmi_host_rabs_stream", "dmi_host_rabs_constant_stream", "dmi_tile_sort_slots", "dmi_decode_attributes", "dmi_decoded_free", "dmi_decode_mesh", "dmi_decoded_mesh_free", "dmi_decode_connectivity", "dmi_decoded_conn_free", "dmi_last_decode_timings", "dmi_free", "dmi_free_many", "dmi_strerror", "dmi_last_error", "dmi_device_count", "dmi_release_cached_memory",
           "dmi_init", "dmi_last_call_timings", "dmi_device_corner_table", "dmi_encode_mesh_device", "dmi_meshes_build", "dmi_built_meshes_prepare", "dmi_last_build_timings", "dmi_device_attribute_table", "dmi_built_meshes_info", "dmi_built_meshes_free", "dmi_thread_host_threads", "dmi_usable_host_threads",
           "dmi_transcoder_create", "dmi_transcoder_reserve", "dmi_transcoder_push", "dmi_transcoder_finish", "dmi_transcoder_result", "dmi_transcoder_timings", "dmi_transcoder_counts", "dmi_transcoder_destroy",
           "dmi_host_alloc", "dmi_host_free", "dmi_host_is_registered",
           "dmi_transcode_assets", "dmi_transcoded_file", "dmi_transcoded_blobs", "dmi_transcoded_stats", "dmi_transcoded_free", "dmi_json_roundtrip", "dmi_transcoded_table", "dmi_transcoded_blocks",
           "dmi_transcoder_stages", "dmi_set_default_debug", "dmi_configure_process"]


def library_path():
    return _LIB


def build_library():
    """hipcc cross-compile of csrc/ for gfx950 (no GPU needed)."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(_PKG, "csrc")])


def load_library():
    """Load libdraco_mi.so; fails loudly if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own HIP / HSA runtime; a process that loads the system runtime first (through libdraco_mi.so) and
    # torch's second ends up with two of them, and the second finds no GPU.  With torch imported first there is one runtime for both.
    if "torch" not in sys.modules and not os.environ.get("DMI_NO_TORCH_PREIMPORT") and "asan" not in os.path.basename(_LIB):
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    if not os.path.exists(_LIB):
        raise ImportError(f"{_LIB} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                          "draco-oxide_amd has no CPU fallback.")
    L = C.CDLL(_LIB)
    L.dmi_strerror.restype = C.c_char_p
    L.dmi_strerror.argtypes = [C.c_int]
    L.dmi_last_error.restype = C.c_char_p
    L.dmi_device_count.restype = C.c_int
    L.dmi_release_cached_memory.restype = None
    L.dmi_free.argtypes = [C.POINTER(_Buffer)]
    L.dmi_free_many.argtypes = [C.POINTER(_Buffer), C.c_uint32]
    L.dmi_free_many.restype = None
    L.dmi_encode_attributes.argtypes = [C.POINTER(_Attribute), C.POINTER(_CornerTable), C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(_Config), C.POINTER(_Buffer)]
    L.dmi_job_create.argtypes = [C.POINTER(_Attribute), C.POINTER(_CornerTable), C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(_Config), C.POINTER(C.c_void_p)]
    L.dmi_job_encode.argtypes = [C.c_void_p, C.POINTER(_Buffer)]
    L.dmi_jobs_encode.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(_Buffer)]
    L.dmi_job_timings.argtypes = [C.c_void_p, C.POINTER(_Timings)]
    L.dmi_last_call_timings.argtypes = [C.POINTER(_Timings)]
    L.dmi_job_destroy.argtypes = [C.c_void_p]
    L.dmi_encode_mesh.argtypes = [C.POINTER(_Mesh), C.POINTER(_Config), C.POINTER(_Buffer)]
    L.dmi_decode_mesh.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_Config), C.POINTER(_DecodedMesh)]
    L.dmi_last_decode_timings.argtypes = [C.POINTER(_DecodeTimings)]
    L.dmi_decoded_mesh_free.argtypes = [C.POINTER(_DecodedMesh)]
    L.dmi_decoded_mesh_free.restype = None
    L.dmi_decode_connectivity.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_Conn), C.POINTER(C.c_size_t)]
    L.dmi_decoded_conn_free.argtypes = [C.POINTER(_Conn)]
    L.dmi_decoded_conn_free.restype = None
    L.dmi_device_corner_table.argtypes = [C.POINTER(_Mesh), C.POINTER(_Config), C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.dmi_encode_mesh_device.argtypes = [C.POINTER(_Mesh), C.POINTER(_Config), C.POINTER(_Buffer)]
    L.dmi_mesh_prepare.argtypes = [C.POINTER(_Mesh), C.POINTER(_Config), C.POINTER(_Buffer), C.POINTER(C.c_void_p)]
    L.dmi_mesh_build.argtypes = [C.POINTER(_RawAttribute), C.c_uint32, C.c_void_p, C.c_uint32, C.POINTER(_BuiltMesh)]
    L.dmi_built_mesh_free.argtypes = [C.POINTER(_BuiltMesh)]
    L.dmi_built_mesh_free.restype = None
    L.dmi_meshes_prepare.argtypes = [C.POINTER(_Mesh), C.c_uint32, C.POINTER(_Config), C.POINTER(_Buffer), C.POINTER(C.c_void_p)]
    L.dmi_encode_connectivity.argtypes = [C.POINTER(_Mesh), C.POINTER(_Buffer), C.POINTER(_Conn)]
    L.dmi_shard_meshes.argtypes = [C.POINTER(_Mesh), C.c_uint32, C.c_uint32, C.c_void_p]
    L.dmi_meshes_prepare_devices.argtypes = [C.POINTER(_Mesh), C.c_uint32, C.POINTER(_Config), C.c_void_p, C.POINTER(_Buffer), C.POINTER(C.c_void_p)]
    L.dmi_jobs_encode_devices.argtypes = [C.POINTER(C.c_void_p), C.c_uint32, C.POINTER(_Buffer)]
    L.dmi_conn_free.argtypes = [C.POINTER(_Conn)]
    L.dmi_decode_attributes.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_CornerTable), C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(_Config), C.POINTER(_Decoded)]
    L.dmi_decoded_free.argtypes = [C.POINTER(_Decoded)]
    L.dmi_decoded_free.restype = None
    L.dmi_host_rans_stream.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint64, C.POINTER(_Buffer)]
    L.dmi_host_rabs_stream.argtypes = [C.c_uint8, C.c_void_p, C.c_uint64, C.POINTER(_Buffer)]
    L.dmi_host_rabs_constant_stream.argtypes = [C.c_uint8, C.c_uint32, C.c_uint64, C.POINTER(_Buffer)]
    L.dmi_tile_sort_slots.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(_Config), C.c_void_p, C.c_void_p]
    L.dmi_init.argtypes = [C.c_int, C.c_size_t, C.c_size_t]
    L.dmi_meshes_build.argtypes = [C.POINTER(_RawMesh), C.c_uint32, C.POINTER(_Config), C.c_uint32, C.POINTER(_BuiltMesh)]
    L.dmi_built_meshes_prepare.argtypes = [C.POINTER(_BuiltMesh), C.c_uint32, C.POINTER(_Config), C.POINTER(_Buffer), C.POINTER(C.c_void_p)]
    L.dmi_transcoder_create.argtypes = [C.POINTER(_Config), C.c_uint64, C.c_uint64, _TRANSCODE_DONE, C.c_void_p]
    L.dmi_transcoder_create.restype = C.c_void_p
    L.dmi_transcoder_reserve.argtypes = [C.c_void_p, C.c_uint32]
    L.dmi_transcoder_push.argtypes = [C.c_void_p, C.POINTER(_RawMesh), C.c_uint32]
    L.dmi_transcoder_finish.argtypes = [C.c_void_p]
    L.dmi_transcoder_result.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_Buffer), C.POINTER(_Buffer), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.dmi_transcoder_timings.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.dmi_transcoder_destroy.argtypes = [C.c_void_p]
    L.dmi_transcoder_destroy.restype = None
    L.dmi_last_build_timings.argtypes = [C.POINTER(_BuildTimings)]
    L.dmi_built_meshes_info.argtypes = [C.POINTER(_BuiltMesh), C.c_uint32, C.c_void_p, C.c_void_p]
    L.dmi_built_meshes_free.argtypes = [C.POINTER(_BuiltMesh), C.c_uint32]
    L.dmi_built_meshes_free.restype = None
    L.dmi_thread_host_threads.argtypes = [C.c_uint32]
    L.dmi_thread_host_threads.restype = None
    L.dmi_usable_host_threads.restype = C.c_int
    L.dmi_device_attribute_table.argtypes = [C.POINTER(_Mesh), C.POINTER(_Config), C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.dmi_host_alloc.argtypes = [C.c_size_t]
    L.dmi_host_alloc.restype = C.c_void_p
    L.dmi_host_free.argtypes = [C.c_void_p]
    L.dmi_host_free.restype = None
    L.dmi_host_is_registered.argtypes = [C.c_void_p, C.c_size_t]
    L.dmi_transcode_assets.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(_Config), C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]
    L.dmi_transcoded_file.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_uint32)]
    L.dmi_transcoded_blobs.argtypes = [C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    L.dmi_transcoded_stats.argtypes = [C.c_void_p, C.POINTER(_TranscodeStats)]
    L.dmi_transcoded_table.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.dmi_transcoded_blocks.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32]
    L.dmi_transcoded_blocks.restype = C.c_uint32
    L.dmi_transcoded_free.argtypes = [C.c_void_p]
    L.dmi_transcoded_free.restype = None
    L.dmi_json_roundtrip.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(_Buffer)]
    L.dmi_set_default_debug.argtypes = [C.POINTER(_Debug)]
    L.dmi_set_default_debug.restype = None
    L.dmi_configure_process.argtypes = [C.POINTER(_ProcessOptions)]
    _lib = L
    _sync_default_debug(L)
    # process options the environment asks for (the library reads none of these itself): DMI_HUGE_PAGE_NEW / DMI_NUMA_PIN opt in, DMI_NO_THP opts out,
    # DMI_HOST_CACHE_MB / DMI_DEVICE_CACHE_MB / DMI_DECODE_BUDGET_MB size the caches and the decoder's allocation budget
    env = os.environ
    def _mb(name):
        try:
            return max(0, int(env.get(name, "0")))
        except ValueError:
            return 0
    if any(k in env for k in ("DMI_HUGE_PAGE_NEW", "DMI_NUMA_PIN", "DMI_NO_THP", "DMI_HOST_CACHE_MB", "DMI_DEVICE_CACHE_MB", "DMI_DECODE_BUDGET_MB")):
        configure_process("DMI_HUGE_PAGE_NEW" in env, "DMI_NUMA_PIN" in env, "DMI_NO_THP" in env, _mb("DMI_HOST_CACHE_MB"), _mb("DMI_DEVICE_CACHE_MB"), _mb("DMI_DECODE_BUDGET_MB"))
    return L


def _check(rc):
    if rc != 0:
        L = load_library()
        raise DracoMiError(rc, L.dmi_strerror(rc).decode(), L.dmi_last_error().decode())


def device_count():
    return int(load_library().dmi_device_count())


def thread_host_threads(n):
    """dmi_thread_host_threads: cap on the host threads of the library calls THIS thread makes (0 = none)."""
    load_library().dmi_thread_host_threads(int(n))


def usable_host_threads():
    return int(load_library().dmi_usable_host_threads())


def release_cached_memory():
    """Hand back what the library keeps between calls (device chunks, pinned staging, recycled host arrays)."""
    load_library().dmi_release_cached_memory()


def _take(buf):
    L = load_library()
    try:
        return C.string_at(buf.data, buf.len) if buf.len else b""
    finally:
        L.dmi_free(C.byref(buf))


# --------------------------------------------------------------------------------------------------
# Data model (host side)
# --------------------------------------------------------------------------------------------------
class Attribute:
    """Unique values + optional point→value map, as core/attribute/mod.rs:26-49 stores them."""

    def __init__(self, values, att_type, domain=DOMAIN_POSITION, unique_id=0, parent_index=-1, point_to_value=None, num_points=None):
        v = np.ascontiguousarray(values)
        if v.ndim == 1:
            v = v.reshape(-1, 1)
        if v.dtype == np.float32:
            self.component_type = F32
        elif v.dtype == np.uint32:
            self.component_type = U32
        elif v.dtype == np.int32:
            self.component_type = I32
        else:
            raise TypeError("attribute values must be float32 / uint32 / int32")
        self.values = v
        self.att_type = att_type
        self.domain = domain
        self.unique_id = unique_id
        self.parent_index = parent_index
        self.point_to_value = None if point_to_value is None else np.ascontiguousarray(point_to_value, dtype=np.uint32)
        self.num_points = int(num_points if num_points is not None else (len(self.point_to_value) if self.point_to_value is not None else v.shape[0]))

    def _c(self):
        a = _Attribute()
        a.values = self.values.ctypes.data
        a.num_unique = self.values.shape[0]
        a.component_type = self.component_type
        a.num_components = self.values.shape[1]
        a.att_type = self.att_type
        a.domain = self.domain
        a.unique_id = self.unique_id
        a.parent_index = self.parent_index
        a.point_to_value = None if self.point_to_value is None else self.point_to_value.ctypes.data
        a.num_points = self.num_points
        return a


class Mesh:
    """faces (point indices) + attributes; attribute 0 must be the Position attribute."""

    def __init__(self, faces, attributes):
        self.faces = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)
        self.attributes = list(attributes)

    def _c(self):
        arr = (_Attribute * len(self.attributes))(*[a._c() for a in self.attributes])
        m = _Mesh()
        m.faces = self.faces.ctypes.data
        m.num_faces = self.faces.shape[0]
        m.atts = arr
        m.num_atts = len(self.attributes)
        m._keep = arr
        return m


def _dedup_rows(rows, want_classes=False):
    """First-occurrence value dedup with f32 `==` semantics (core/attribute/mod.rs:394-452): -0.0 == 0.0, NaN != NaN.
    want_classes: also the BYTE CLASS of every row — what the point merge compares (core/mesh/builder.rs:254-279 hashes the bytes of
    the row's unique value): the first occurrence of the row's `==` class, for a NaN row the first row with the same bytes."""
    r = np.ascontiguousarray(rows)
    if r.shape[0] == 0:
        return (r, None, np.zeros(0, np.int64)) if want_classes else (r, None)
    key = r
    nan_rows = None
    if r.dtype == np.float32:
        key = r + np.float32(0.0)          # -0.0 → +0.0
        key = np.where(key == 0, np.float32(0.0), key)
        nan_rows = np.isnan(r).any(axis=1)
    kb = np.ascontiguousarray(key).view(np.dtype((np.void, key.dtype.itemsize * key.shape[1]))).ravel()
    _, first, inv = np.unique(kb, return_index=True, return_inverse=True)
    rep = first[inv]                         # first occurrence of each row's class
    classes = rep
    if nan_rows is not None and nan_rows.any():
        idx = np.arange(r.shape[0])
        raw = r.view(np.dtype((np.void, r.dtype.itemsize * r.shape[1]))).ravel()
        _, rfirst, rinv = np.unique(raw, return_index=True, return_inverse=True)
        classes = np.where(nan_rows, rfirst[rinv], rep)   # byte-identical NaN rows: one class (a NaN row's bytes never equal a NaN-free row's)
        rep = np.where(nan_rows, idx, rep)   # NaN rows are never merged as VALUES
    is_first = rep == np.arange(r.shape[0])
    if is_first.all():
        return (r, None, classes) if want_classes else (r, None)
    new_id = np.cumsum(is_first) - 1
    p2v = new_id[rep].astype(np.uint32)
    vals = np.ascontiguousarray(r[is_first])
    return (vals, p2v, classes) if want_classes else (vals, p2v)


class MeshBuilder:
    """core/mesh/builder.rs:14-90: per-attribute value dedup (Attribute::from), Position swapped to slot 0, points that
    agree in every attribute merged, degenerate faces dropped, unreferenced points removed."""

    def __init__(self):
        self._atts = []
        self._faces = None

    def add_attribute(self, data, att_type, domain=DOMAIN_POSITION, parents=()):
        d = np.ascontiguousarray(data)
        if d.ndim == 1:
            d = d.reshape(-1, 1)
        self._atts.append(dict(data=d, type=att_type, domain=domain, parents=list(parents), id=len(self._atts)))
        return len(self._atts) - 1

    def set_connectivity_attribute(self, faces):
        self._faces = np.ascontiguousarray(faces, dtype=np.uint32).reshape(-1, 3)

    def build(self):
        """MeshBuilder::build through the library (dmi_mesh_build, C++, host only)."""
        L = load_library()
        n = len(self._atts)
        raw = (_RawAttribute * max(n, 1))()
        keep = []
        np_of = {F32: np.float32, U32: np.uint32, I32: np.int32}
        for i, a in enumerate(self._atts):
            d = a["data"]
            ct = {np.dtype(np.float32): F32, np.dtype(np.uint32): U32, np.dtype(np.int32): I32}.get(d.dtype)
            if ct is None:
                raise TypeError("attribute rows must be float32 / uint32 / int32")
            par = np.ascontiguousarray(a["parents"], dtype=np.uint32)
            keep.append((d, par))
            raw[i].data = d.ctypes.data
            raw[i].num_points = d.shape[0]
            raw[i].component_type, raw[i].num_components, raw[i].att_type, raw[i].domain = ct, d.shape[1], a["type"], a["domain"]
            raw[i].num_parents = len(par)
            raw[i].parents = par.ctypes.data if len(par) else None
        faces = self._faces if self._faces is not None else np.zeros((0, 3), np.uint32)
        built = _BuiltMesh()
        _sync_default_debug(L)
        _check(L.dmi_mesh_build(raw, n, faces.ctypes.data if faces.size else None, faces.shape[0], C.byref(built)))
        try:
            m = built.mesh
            out_faces = np.ctypeslib.as_array(C.cast(m.faces, C.POINTER(C.c_uint32)), shape=(m.num_faces, 3)).copy() if m.num_faces else np.zeros((0, 3), np.uint32)
            out = []
            for i in range(m.num_atts):
                a = m.atts[i]
                dt = np_of[a.component_type]
                vals = (np.frombuffer(C.string_at(a.values, a.num_unique * a.num_components * 4), dtype=dt).reshape(a.num_unique, a.num_components).copy()
                        if a.num_unique else np.zeros((0, a.num_components), dt))
                p2v = np.frombuffer(C.string_at(a.point_to_value, a.num_points * 4), dtype=np.uint32).copy() if a.point_to_value else None
                out.append(Attribute(vals, a.att_type, a.domain, unique_id=a.unique_id, parent_index=a.parent_index, point_to_value=p2v, num_points=a.num_points))
        finally:
            L.dmi_built_mesh_free(C.byref(built))
        return Mesh(out_faces, out)

    def build_numpy(self):
        """The same result with numpy (kept as an independent cross-check of the C++ builder; it does not remove
        unreferenced points, builder.rs:129-189, which callers that index every point never have)."""
        atts = []
        for a in self._atts:
            vals, p2v, classes = _dedup_rows(a["data"], want_classes=True)
            atts.append(dict(a, values=vals, p2v=p2v, classes=classes, npoints=a["data"].shape[0]))
        for i, a in enumerate(atts):
            if a["type"] == ATT_POSITION:
                atts[0], atts[i] = atts[i], atts[0]
                break
        faces = self._faces
        npts = int(faces.max()) + 1 if faces.size else 0
        keys = np.stack([a["classes"][:npts].astype(np.uint32) for a in atts], axis=1)   # byte classes, not value ids: NaN rows (see _dedup_rows)
        kb = np.ascontiguousarray(keys).view(np.dtype((np.void, 4 * keys.shape[1]))).ravel()
        _, first, inv = np.unique(kb, return_index=True, return_inverse=True)
        rep = first[inv]
        is_first = rep == np.arange(npts)
        if not is_first.all():
            new_id = (np.cumsum(is_first) - 1).astype(np.uint32)
            mapping = new_id[rep]
            faces = mapping[faces]
            for a in atts:
                m = a["p2v"] if a["p2v"] is not None else np.arange(npts, dtype=np.uint32)
                kept = m[:npts][is_first]
                used = np.zeros(len(a["values"]), bool)
                used[kept] = True
                if not used.all():       # a NaN row's value whose only point was merged away leaves the buffer (Attribute::remove, mod.rs:454-483)
                    kept = (np.cumsum(used) - 1).astype(np.uint32)[kept]
                    a["values"] = np.ascontiguousarray(a["values"][used])
                a["p2v"] = np.ascontiguousarray(kept) if a["p2v"] is not None else None
                a["npoints"] = int(is_first.sum())
        keep = (faces[:, 0] != faces[:, 1]) & (faces[:, 1] != faces[:, 2]) & (faces[:, 2] != faces[:, 0])
        faces = faces[keep]
        id_to_index = {a["id"]: i for i, a in enumerate(atts)}
        out = []
        for a in atts:
            parent = id_to_index[a["parents"][0]] if a["parents"] else -1
            out.append(Attribute(a["values"], a["type"], a["domain"], unique_id=a["id"], parent_index=parent, point_to_value=a["p2v"], num_points=a["npoints"]))
        return Mesh(faces, out)


BUILD_HOST_VALUES = 1
_NP_CT = {np.dtype(np.float32): F32, np.dtype(np.uint32): U32, np.dtype(np.int32): I32}
_CT_NP = {F32: np.float32, U32: np.uint32, I32: np.int32}
_INDEX_CT = {np.dtype(np.uint8): 1, np.dtype(np.uint16): 3, np.dtype(np.uint32): 5}


class RawMesh:
    """What MeshBuilder collects for one primitive, WITHOUT copying: views of the accessors' bytes (io/gltf/decode.rs:2328-2525 hands
    these to MeshBuilder::add_attribute).  `rows`: 2-D array of float32 / uint32 / int32 whose rows may be strided (a glTF bufferView
    with a byteStride); `indices`: contiguous uint8 / uint16 / uint32, 3 per face."""

    def __init__(self):
        self.atts = []
        self.indices = None

    def add_attribute(self, rows, att_type, domain=DOMAIN_POSITION, parents=()):
        r = rows if rows.ndim == 2 else rows.reshape(-1, 1)
        if r.dtype not in _NP_CT or (r.shape[1] > 1 and r.strides[1] != 4):
            raise TypeError("attribute rows must be float32 / uint32 / int32 with contiguous components")
        # the library reads rows `byte_stride` apart as an unsigned distance: a reversed view (negative stride), a broadcast one (0) or rows that
        # overlap (stride below the row size) are copied into rows of their own first
        if r.shape[0] > 1 and not (4 * r.shape[1] <= r.strides[0] < (1 << 31)):
            r = np.ascontiguousarray(r)
        self.atts.append((r, att_type, domain, np.ascontiguousarray(parents, dtype=np.uint32)))
        return len(self.atts) - 1

    def set_indices(self, idx):
        i = np.ascontiguousarray(idx).reshape(-1)
        if i.dtype not in _INDEX_CT:
            i = i.astype(np.uint32)
        self.indices = i[: len(i) // 3 * 3]


class BuiltBatch:
    """The meshes of one dmi_meshes_build call (library-owned; `free()` / context manager releases them).  `mesh(j)` copies mesh j into
    a host `Mesh` (needs host_values=True at build time); `summary(j)` = (num_faces, num_points, [unique ids in slot order])."""

    def __init__(self, arr, n, keep):
        self._arr, self._n, self._keep = arr, n, keep

    def __len__(self):
        return self._n

    def num_faces(self, j):
        return int(self._arr[j].mesh.num_faces)

    def summary(self, j):
        m = self._arr[j].mesh
        return int(m.num_faces), int(m.atts[0].num_points) if m.num_atts else 0, [int(m.atts[i].unique_id) for i in range(m.num_atts)]

    def mesh(self, j):
        m = self._arr[j].mesh
        faces = np.ctypeslib.as_array(C.cast(m.faces, C.POINTER(C.c_uint32)), shape=(m.num_faces, 3)).copy() if m.num_faces else np.zeros((0, 3), np.uint32)
        out = []
        for i in range(m.num_atts):
            a = m.atts[i]
            if a.num_unique and not a.values:
                raise ValueError("the values of this batch stayed on the device (build with host_values=True to copy meshes out)")
            dt = _CT_NP[a.component_type]
            vals = (np.frombuffer(C.string_at(a.values, a.num_unique * a.num_components * 4), dtype=dt).reshape(a.num_unique, a.num_components).copy()
                    if a.num_unique else np.zeros((0, a.num_components), dt))
            p2v = np.frombuffer(C.string_at(a.point_to_value, a.num_points * 4), dtype=np.uint32).copy() if a.point_to_value else None
            out.append(Attribute(vals, a.att_type, a.domain, unique_id=a.unique_id, parent_index=a.parent_index, point_to_value=p2v, num_points=a.num_points))
        return Mesh(faces, out)

    def counts(self):
        """(num_faces, num_points) of every mesh as uint32 arrays — one library call."""
        nf, npts = np.zeros(max(self._n, 1), np.uint32), np.zeros(max(self._n, 1), np.uint32)
        if self._n:
            _check(load_library().dmi_built_meshes_info(self._arr, self._n, nf.ctypes.data, npts.ctypes.data))
        return nf[: self._n], npts[: self._n]

    def free(self):
        if self._n:
            load_library().dmi_built_meshes_free(self._arr, self._n)
        self._n = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.free()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def _raw_mesh_array(raw_meshes):
    """RawMesh list → (dmi_raw_mesh array, what must stay alive while the library reads it)."""
    n = len(raw_meshes)
    arr = (_RawMesh * n)()
    keep = []
    for j, rm in enumerate(raw_meshes):
        acc = (_RawAccessor * max(len(rm.atts), 1))()
        for i, (r, att_type, domain, par) in enumerate(rm.atts):
            acc[i].data = r.ctypes.data
            acc[i].count = r.shape[0]
            acc[i].byte_stride = r.strides[0] if r.shape[0] > 1 else 0
            acc[i].component_type, acc[i].num_components, acc[i].att_type, acc[i].domain = _NP_CT[r.dtype], r.shape[1], att_type, domain
            acc[i].num_parents = len(par)
            acc[i].parents = par.ctypes.data if len(par) else None
        idx = rm.indices if rm.indices is not None else np.zeros(0, np.uint32)
        arr[j].atts, arr[j].n_atts = acc, len(rm.atts)
        arr[j].indices = idx.ctypes.data if len(idx) else None
        arr[j].index_type = _INDEX_CT[idx.dtype]
        arr[j].num_faces = len(idx) // 3
        keep.append((acc, idx, rm))
    return arr, keep


def meshes_build(raw_meshes, cfg=None, host_values=False):
    """dmi_meshes_build: MeshBuilder::build for a list of RawMesh on the device → BuiltBatch."""
    L = load_library()
    cfg = cfg or Config.default()
    n = len(raw_meshes)
    if n == 0:
        return BuiltBatch(None, 0, None)
    arr, keep = _raw_mesh_array(raw_meshes)
    built = (_BuiltMesh * n)()
    c = cfg._c()
    _check(L.dmi_meshes_build(arr, n, C.byref(c), BUILD_HOST_VALUES if host_values else 0, built))
    return BuiltBatch(built, n, keep)


def _address_of(buf):
    """(address, length) of a bytes-like object's memory (bytes / bytearray / memoryview / numpy array), without copying."""
    a = np.frombuffer(buf, dtype=np.uint8) if not isinstance(buf, np.ndarray) else buf
    return a.ctypes.data, a.nbytes


class HostBuffer:
    """dmi_host_alloc memory as a writable uint8 numpy array (`.array`): what an importer reads a file INTO — accessors inside it go up to the device
    where they lie (no host pack, no staging copy).  `HostBuffer.holding(data)`: a buffer with a copy of `data` in it."""

    @classmethod
    def holding(cls, data):
        a = np.frombuffer(data, dtype=np.uint8)
        hb = cls(max(1, a.nbytes))
        hb.array[: a.nbytes] = a
        hb.nbytes = a.nbytes
        return hb

    def view(self):
        """The bytes put in by holding() as a memoryview of the page-locked memory."""
        return memoryview(self.array[: getattr(self, "nbytes", len(self.array))])

    def __init__(self, nbytes):
        self._p = load_library().dmi_host_alloc(nbytes)
        if not self._p:
            raise MemoryError("dmi_host_alloc")
        self.array = np.ctypeslib.as_array(C.cast(self._p, C.POINTER(C.c_uint8)), shape=(nbytes,))

    def free(self):
        if self._p:
            self.array = None
            load_library().dmi_host_free(self._p)
            self._p = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def json_roundtrip(text):
    """dmi_json_roundtrip: the library's JSON layer (parse + compact write) on `text` (bytes) → bytes."""
    L = load_library()
    out = _Buffer()
    _check(L.dmi_json_roundtrip(text, len(text), C.byref(out)))
    try:
        return C.string_at(out.data, out.len)
    finally:
        L.dmi_free(C.byref(out))


class _TranscodedHandle:
    """Owner of a dmi_transcoded: every view handed out keeps it alive; the library's memory goes back when the last one is gone."""

    def __init__(self, h):
        self.h = h

    def __del__(self):
        try:
            if self.h:
                load_library().dmi_transcoded_free(self.h)
                self.h = None
        except Exception:
            pass


def _fast_address(buf):
    """(address, length) of a bytes-like object without going through numpy for the common types."""
    if type(buf) is bytes:
        return C.cast(C.c_char_p(buf), C.c_void_p).value or 0, len(buf)
    return _address_of(buf)


class AssetList:
    """A list of assets marshalled for dmi_transcode_assets once (the C array of dmi_gltf_asset + whatever keeps the bytes alive): a caller that
    transcodes the same list again — or wants the marshalling out of a timed region, like DeviceMesh for a mesh — passes this instead of the list."""

    def __init__(self, assets):
        n = len(assets)
        self.n = n
        self.keep = []
        tab = np.zeros((max(n, 1), 6), np.uint64)          # dmi_gltf_asset: glb, glb_bytes, json, json_bytes, buffers, n_buffers (48 bytes)
        assert C.sizeof(_GltfAsset) == 48
        glb_addr, glb_len = [0] * n, [0] * n
        for i, a in enumerate(assets):
            if isinstance(a, tuple):
                js, buffers = a
                js = bytes(js)
                spans = (_Span * max(len(buffers), 1))()
                for k, b in enumerate(buffers):
                    addr, nb = _fast_address(b) if len(b) else (0, 0)
                    spans[k].data, spans[k].bytes = addr, nb
                self.keep.append((js, buffers, spans))
                tab[i, 2], tab[i, 3] = _fast_address(js)
                tab[i, 4], tab[i, 5] = C.addressof(spans), len(buffers)
            else:
                glb_addr[i], glb_len[i] = _fast_address(a)
                self.keep.append(a)
        if n:
            tab[:n, 0], tab[:n, 1] = glb_addr, glb_len
        self.table = tab

    def __len__(self):
        return self.n


def transcode_assets(assets, cfg=None, devices=None):
    """dmi_transcode_assets: `assets` = GLB containers (bytes-like) or (json_bytes, [buffer bytes-like, ...]) pairs — or an AssetList of them →
    ([(glb, [blob, ...]), ...], stats).  glb and the blobs are memoryviews of memory the library owns (slices of the result's arena blocks; the blobs lie
    inside their file; everything is released when the last view is gone); stats = dmi_transcode_stats as a dict.  devices: HIP ordinals (one
    dmi_transcoder each; None: cfg.device)."""
    L = load_library()
    cfg = cfg or Config.default()
    al = assets if isinstance(assets, AssetList) else AssetList(assets)
    n = len(al)
    dev = None
    if devices:
        dev = (C.c_int32 * len(devices))(*[int(d) for d in devices])
    c = cfg._c()
    h = C.c_void_p()
    _check(L.dmi_transcode_assets(C.c_void_p(al.table.ctypes.data), n, C.byref(c), dev, len(devices) if devices else 0, 0, C.byref(h)))
    owner = _TranscodedHandle(h.value)
    st = _TranscodeStats()
    _check(L.dmi_transcoded_stats(h, C.byref(st)))
    addr, size, nblob = np.zeros(n, np.uint64), np.zeros(n, np.uint64), np.zeros(n, np.uint32)
    cap = max(1, st.primitives)
    offs, sizes = np.zeros(cap, np.uint64), np.zeros(cap, np.uint64)
    _check(L.dmi_transcoded_table(h, addr.ctypes.data, size.ctypes.data, nblob.ctypes.data, offs.ctypes.data, sizes.ctypes.data, cap))
    # the arena's blocks, wrapped once each; a file is a slice of its block (a thousand ctypes array types and from_address calls were 3 ms per call)
    nb = L.dmi_transcoded_blocks(h, None, None, 0)
    baddr, bbytes = np.zeros(max(nb, 1), np.uint64), np.zeros(max(nb, 1), np.uint64)
    L.dmi_transcoded_blocks(h, baddr.ctypes.data, bbytes.ctypes.data, nb)
    blocks = []
    for k in range(nb):
        raw = (C.c_uint8 * int(bbytes[k])).from_address(int(baddr[k]))
        raw._owner = owner
        blocks.append((int(baddr[k]), int(baddr[k]) + int(bbytes[k]), memoryview(raw).cast("B")))
    blocks.sort()
    starts = [b[0] for b in blocks]
    import bisect
    out = []
    at = 0
    addr, size, nblob, offs, ends = addr.tolist(), size.tolist(), nblob.tolist(), offs.tolist(), (offs + sizes).tolist()
    empty = memoryview(b"")
    for i in range(n):
        if size[i]:
            lo, hi, view = blocks[bisect.bisect_right(starts, addr[i]) - 1]
            glb = view[addr[i] - lo: addr[i] - lo + size[i]]
        else:
            glb = empty
        k = nblob[i]
        out.append((glb, [glb[offs[q]: ends[q]] for q in range(at, at + k)]))
        at += k
    return out, {name: getattr(st, name) for name, _ in _TranscodeStats._fields_ if name != "pad"}


def last_build_timings():
    t = _BuildTimings()
    _check(load_library().dmi_last_build_timings(C.byref(t)))
    return {k: getattr(t, k) for k, _ in _BuildTimings._fields_}


def built_meshes_prepare(batch, which=None, cfg=None):
    """dmi_built_meshes_prepare: the connectivity stage + job creation for meshes a BuiltBatch holds on the device (all of them, or the
    indices `which`) — nothing is uploaded again.  Returns the Jobs in that order."""
    L = load_library()
    cfg = cfg or Config.default()
    which = list(range(len(batch))) if which is None else list(which)
    n = len(which)
    if n == 0:
        return []
    if n == len(batch) and which == list(range(n)):
        arr = batch._arr
    else:
        arr = (_BuiltMesh * n)()
        for k, j in enumerate(which):
            arr[k] = batch._arr[j]
    c = cfg._c()
    heads = (_Buffer * n)()
    handles = (C.c_void_p * n)()
    _check(L.dmi_built_meshes_prepare(arr, n, C.byref(c), heads, handles))
    return [Job(handles[i], _take(heads[i])) for i in range(n)]


class Transcoder:
    """dmi_transcoder: the transcoder's per-primitive loop inside the library.  push(RawMesh list) as the importer produces them; stages run build →
    prepare → encode on library threads; on_done(first, count) is called (from a library thread) when the primitives [first, first + count) are
    final; result(i) → ((header + connectivity, attribute section) as zero-copy uint8 views, num_faces, num_points) or None (no face left).  The
    views are valid until close()."""

    def __init__(self, cfg=None, expected_triangles=0, n_primitives=0, on_done=None, stage_triangles=0):
        L = load_library()
        self._L = L
        self.errors = []

        def _cb(user, first, count):
            try:
                on_done(first, count)
            except BaseException as e:  # noqa: BLE001 — nothing may propagate into the library's thread; finish() raises it
                self.errors.append(e)

        self._cb = _TRANSCODE_DONE(_cb) if on_done else C.cast(None, _TRANSCODE_DONE)
        c = (cfg or Config.default())._c()
        L.dmi_transcoder_create.restype = C.c_void_p
        self._h = C.c_void_p(L.dmi_transcoder_create(C.byref(c), C.c_uint64(int(expected_triangles)), C.c_uint64(int(stage_triangles)), self._cb, None))
        if not self._h:
            raise MemoryError("dmi_transcoder_create")
        self._keep = []
        if n_primitives:
            _check(L.dmi_transcoder_reserve(self._h, int(n_primitives)))

    def reserve(self, n_primitives):
        """A hint: about this many primitives in all."""
        _check(self._L.dmi_transcoder_reserve(self._h, int(n_primitives)))

    def push(self, raw_meshes):
        if not raw_meshes:
            return
        arr, keep = _raw_mesh_array(raw_meshes)
        self._keep.append(keep)                                            # (the arrays the descriptors point to: alive until close())
        _check(self._L.dmi_transcoder_push(self._h, arr, len(raw_meshes)))

    def finish(self):
        _check(self._L.dmi_transcoder_finish(self._h))
        if self.errors:
            raise self.errors[0]

    def result(self, i):
        head, sec = _Buffer(), _Buffer()
        nf, npts = C.c_uint32(0), C.c_uint32(0)
        _check(self._L.dmi_transcoder_result(self._h, int(i), C.byref(head), C.byref(sec), C.byref(nf), C.byref(npts)))
        if not nf.value:
            return None
        view = lambda b: np.ctypeslib.as_array((C.c_uint8 * b.len).from_address(b.data)) if b.len else np.zeros(0, np.uint8)  # noqa: E731
        return (view(head), view(sec)), int(nf.value), int(npts.value)

    def timings(self):
        b, p, e = C.c_double(0), C.c_double(0), C.c_double(0)
        _check(self._L.dmi_transcoder_timings(self._h, C.byref(b), C.byref(p), C.byref(e)))
        return {"build_s": b.value * 1e-3, "prepare_s": p.value * 1e-3, "encode_s": e.value * 1e-3}

    def close(self):
        if self._h:
            self._L.dmi_transcoder_destroy(self._h)
            self._h = None
        self._keep = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Config:
    """encode::Config — only `Config.default()` is constructible in the reference (encode/mod.rs:22-42).
    The extra fields select the internal variants the reference compiles but does not expose."""

    def __init__(self, pos_bits=0, uv_bits=0, generic_bits=0, pos_scheme=0, device=0, stream=None, flags=0, debug=None):
        self.pos_bits, self.uv_bits, self.generic_bits, self.pos_scheme = pos_bits, uv_bits, generic_bits, pos_scheme
        self.device, self.stream, self.flags = device, stream, flags
        self.debug = debug

    @classmethod
    def default(cls):
        return cls()

    def _c(self):
        c = _Config()
        c.pos_bits, c.uv_bits, c.generic_bits, c.pos_scheme = self.pos_bits, self.uv_bits, self.generic_bits, self.pos_scheme
        c.device = self.device
        c.stream = self.stream
        c.flags = self.flags
        # the typed switches: this Config's own (`debug`: a _Debug, e.g. debug_from_env({...})) or what the environment asks for right now
        dbg = self.debug if self.debug is not None else debug_from_env()
        c._debug_keep = dbg                      # (the struct must outlive the call that reads the pointer)
        c.debug = C.pointer(dbg)
        if _lib is not None:
            _lib.dmi_set_default_debug(C.byref(dbg))   # entry points without a config (job.encode of …, host stages) see the same switches
        return c


# --------------------------------------------------------------------------------------------------
# Entry points
# --------------------------------------------------------------------------------------------------
def encode_mesh(mesh, cfg=None):
    """Whole .drc for `mesh` (header + host Edgebreaker connectivity + device attribute section)."""
    L = load_library()
    cfg = cfg or Config.default()
    m, c, out = mesh._c(), cfg._c(), _Buffer()
    _check(L.dmi_encode_mesh(C.byref(m), C.byref(c), C.byref(out)))
    return _take(out)


def encode(mesh, writer, cfg=None):
    """encode::encode(mesh, &mut writer, cfg): appends the .drc bytes to `writer` (a bytearray)."""
    writer.extend(encode_mesh(mesh, cfg))


class Connectivity:
    """Host connectivity stage output: header+connectivity bytes and the flat tables/sequences/seeds."""

    def __init__(self, mesh):
        L = load_library()
        self._L = L
        self._conn = _Conn()
        m, out = mesh._c(), _Buffer()
        _sync_default_debug(L)
        _check(L.dmi_encode_connectivity(C.byref(m), C.byref(out), C.byref(self._conn)))
        self.bytes = _take(out)
        self.num_tables = self._conn.num_tables
        self.num_seeds = self._conn.num_seeds

    def _arr(self, ptr, n):
        if not ptr or n == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n,)).copy()

    def seeds(self):
        return self._arr(self._conn.seeds, self._conn.num_seeds)

    def table(self, i):
        t = self._conn.tables[i]
        nc = 3 * t.num_faces
        return dict(num_faces=t.num_faces, num_vertices=t.num_vertices, corner_to_point=self._arr(t.corner_to_point, nc),
                    corner_to_vertex=self._arr(t.corner_to_vertex, nc), opposite=self._arr(t.opposite, nc),
                    left_most_corner=self._arr(t.left_most_corner, t.num_vertices), sequence=self._arr(t.sequence, t.sequence_len))

    def close(self):
        if self._conn.owner:
            self._L.dmi_conn_free(C.byref(self._conn))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _tables_c(tables):
    keep = []
    arr = (_CornerTable * len(tables))()
    for i, t in enumerate(tables):
        ct = arr[i]
        c2p = np.ascontiguousarray(t["corner_to_point"], dtype=np.uint32)
        c2v = np.ascontiguousarray(t["corner_to_vertex"], dtype=np.uint32)
        opp = np.ascontiguousarray(t["opposite"], dtype=np.uint32)
        keep += [c2p, c2v, opp]
        ct.num_faces = len(c2p) // 3
        ct.num_vertices = int(t["num_vertices"])
        ct.corner_to_point, ct.corner_to_vertex, ct.opposite = c2p.ctypes.data, c2v.ctypes.data, opp.ctypes.data
        if t.get("left_most_corner") is not None:
            lmc = np.ascontiguousarray(t["left_most_corner"], dtype=np.uint32)
            keep.append(lmc)
            ct.left_most_corner = lmc.ctypes.data
        if t.get("sequence") is not None:
            seq = np.ascontiguousarray(t["sequence"], dtype=np.uint32)
            keep.append(seq)
            ct.sequence = seq.ctypes.data
            ct.sequence_len = len(seq)
    return arr, keep


def encode_attributes(attributes, tables, seeds=None, cfg=None):
    """attribute::encode_attributes drop-in: attribute-section bytes for attributes[i] on tables[i]."""
    L = load_library()
    cfg = cfg or Config.default()
    atts = (_Attribute * len(attributes))(*[a._c() for a in attributes])
    tabs, keep = _tables_c(tables)
    sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32)
    out, c = _Buffer(), cfg._c()
    _check(L.dmi_encode_attributes(atts, tabs, len(attributes), None if sd is None else sd.ctypes.data, 0 if sd is None else len(sd), C.byref(c), C.byref(out)))
    return _take(out)


def encode_attributes_batch(items, cfg=None):
    """dmi_encode_attributes_batch: items = [(attributes, tables, seeds or None), ...] → one attribute section per item."""
    L = load_library()
    cfg = cfg or Config.default()
    n = len(items)
    arr = (_BatchItem * n)()
    keep = []
    for i, (attributes, tables, seeds) in enumerate(items):
        atts = (_Attribute * len(attributes))(*[a._c() for a in attributes])
        tabs, k = _tables_c(tables)
        sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32)
        keep.append((atts, tabs, k, sd, attributes))
        arr[i].atts, arr[i].tables, arr[i].n_atts = atts, tabs, len(attributes)
        arr[i].seeds = None if sd is None else sd.ctypes.data
        arr[i].n_seeds = 0 if sd is None else len(sd)
    outs = (_Buffer * n)()
    c = cfg._c()
    _check(L.dmi_encode_attributes_batch(arr, n, C.byref(c), outs))
    return [_take(outs[i]) for i in range(n)]


class Job:
    """Resident attribute-encoding job: inputs live in HBM, `encode()` runs the device pipeline."""

    def __init__(self, handle, head=b""):
        self._L = load_library()
        self._h = C.c_void_p(handle)
        self.header_and_connectivity = head

    @classmethod
    def from_tables(cls, attributes, tables, seeds=None, cfg=None):
        L = load_library()
        cfg = cfg or Config.default()
        atts = (_Attribute * len(attributes))(*[a._c() for a in attributes])
        tabs, keep = _tables_c(tables)
        sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32)
        h, c = C.c_void_p(), cfg._c()
        _check(L.dmi_job_create(atts, tabs, len(attributes), None if sd is None else sd.ctypes.data, 0 if sd is None else len(sd), C.byref(c), C.byref(h)))
        return cls(h.value)

    def encode(self):
        out = _Buffer()
        _check(self._L.dmi_job_encode(self._h, C.byref(out)))
        return _take(out)

    def encode_raw(self):
        """encode() at the cost of the C-ABI call alone: the attribute section stays in the library-owned buffer (an EncodedBatch of
        one item: `len(batch[0])`-free access through `.nbytes`, `batch[0]` copies, `.free()` / context manager releases)."""
        outs = (_Buffer * 1)()
        _check(self._L.dmi_job_encode(self._h, outs))
        return EncodedBatch(outs, 1)

    def timings(self):
        t = _Timings()
        _check(self._L.dmi_job_timings(self._h, C.byref(t)))
        return {k: getattr(t, k) for k, _ in _Timings._fields_}

    def close(self):
        if self._h:
            self._L.dmi_job_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def jobs_encode(jobs):
    """Batch form: all jobs (created with the same Config.stream) are encoded together — one host wait for all
    histograms and ONE launch holding every rANS/rABS stream.  Returns the list of attribute sections."""
    L = load_library()
    n = len(jobs)
    handles = (C.c_void_p * n)(*[j._h for j in jobs])
    outs = (_Buffer * n)()
    _check(L.dmi_jobs_encode(handles, n, outs))
    return [_take(outs[i]) for i in range(n)]


class EncodedBatch:
    """The library-owned outputs of one dmi_jobs_encode call (no copy into Python objects): `len(batch)`, `batch.nbytes`,
    `batch[i]` → bytes (copies that item), `batch.free()` / context manager → dmi_free of every buffer."""
    def __init__(self, outs, n):
        self._outs, self._n = outs, n
    def __len__(self):
        return self._n
    @property
    def nbytes(self):
        return sum(self._outs[i].len for i in range(self._n))
    def __getitem__(self, i):
        b = self._outs[i]
        return C.string_at(b.data, b.len) if b.len else b""
    def view(self, i):
        """Item i as a zero-copy uint8 numpy view of the library-owned buffer (valid until free())."""
        b = self._outs[i]
        if not b.len:
            return np.zeros(0, np.uint8)
        return np.ctypeslib.as_array((C.c_uint8 * b.len).from_address(b.data))
    def free(self):
        if self._n:
            load_library().dmi_free_many(self._outs, self._n)
        self._n = 0
    def __enter__(self):
        return self
    def __exit__(self, *a):
        self.free()


def jobs_encode_raw(jobs):
    """jobs_encode at the cost of the C-ABI call alone: returns an EncodedBatch (caller frees)."""
    L = load_library()
    n = len(jobs)
    handles = (C.c_void_p * n)(*[j._h for j in jobs])
    outs = (_Buffer * n)()
    _check(L.dmi_jobs_encode(handles, n, outs))
    return EncodedBatch(outs, n)


def mesh_prepare(mesh, cfg=None):
    """Host stages (corner tables, Edgebreaker, sequencer) + upload; returns a resident Job."""
    L = load_library()
    cfg = cfg or Config.default()
    m, c, head, h = mesh._c(), cfg._c(), _Buffer(), C.c_void_p()
    _check(L.dmi_mesh_prepare(C.byref(m), C.byref(c), C.byref(head), C.byref(h)))
    return Job(h.value, _take(head))


def meshes_prepare(meshes, cfg=None):
    """mesh_prepare for a list of meshes, host stages on a thread pool inside the library (dmi_meshes_prepare)."""
    L = load_library()
    cfg = cfg or Config.default()
    n = len(meshes)
    if n == 0:
        return []
    arr = (_Mesh * n)()
    keep = []
    for i, m in enumerate(meshes):
        cm = m._c()
        keep.append(cm)
        arr[i] = cm
    c = cfg._c()
    heads = (_Buffer * n)()
    handles = (C.c_void_p * n)()
    _check(L.dmi_meshes_prepare(arr, n, C.byref(c), heads, handles))
    return [Job(handles[i], _take(heads[i])) for i in range(n)]


def init(device=0, staging_bytes=0, device_bytes=0):
    """dmi_init: pay the process's one-time costs for `device` now (context, code objects, stream, optional staging / device pool)."""
    L = load_library()
    _check(L.dmi_init(device, staging_bytes, device_bytes))


def last_call_timings():
    """dmi_last_call_timings: stage times of this thread's last whole-mesh / boundary call."""
    t = _Timings()
    _check(load_library().dmi_last_call_timings(C.byref(t)))
    return {k: getattr(t, k) for k, _ in _Timings._fields_}


class _HipArray:
    """A host array copied into hipMalloc'd memory through the HIP runtime the library itself is bound to (DeviceMesh.upload in a process without torch)."""
    _hip = None

    def __init__(self, arr, device):
        if _HipArray._hip is None:
            _HipArray._hip = C.CDLL("libamdhip64.so.7")   # (already loaded as libdraco_mi.so's dependency: the same runtime)
        hip = _HipArray._hip
        arr = np.ascontiguousarray(arr)
        self._p = C.c_void_p()
        if hip.hipSetDevice(int(device)) or hip.hipMalloc(C.byref(self._p), C.c_size_t(max(arr.nbytes, 4))):
            raise DracoMiError(1, "hipMalloc failed")
        if arr.nbytes and hip.hipMemcpy(self._p, C.c_void_p(arr.ctypes.data), C.c_size_t(arr.nbytes), 1):
            raise DracoMiError(1, "hipMemcpy failed")

    def data_ptr(self):
        return self._p.value

    def __del__(self):
        if self._p and _HipArray._hip is not None:
            _HipArray._hip.hipFree(self._p)
            self._p = None


class DeviceMesh:
    """A mesh whose faces, attribute values and point → value maps live in device memory (torch tensors on one HIP device):
    the input of dmi_encode_mesh_device.  Built from a host `Mesh` by `DeviceMesh.upload(mesh, device)`."""

    def __init__(self, mesh, tensors, device):
        self.mesh, self._t, self.device = mesh, tensors, device

    @classmethod
    def upload(cls, mesh, device=0):
        if "torch" not in sys.modules:   # a process without torch (the library on the system's HIP runtime): plain hipMalloc'd arrays
            load_library()
            t = {"faces": _HipArray(mesh.faces.view(np.int32), device), "values": [], "maps": {}}
            for a in mesh.attributes:
                t["values"].append(_HipArray(np.ascontiguousarray(a.values).view(np.int32), device))
                if a.point_to_value is not None and id(a.point_to_value) not in t["maps"]:
                    t["maps"][id(a.point_to_value)] = _HipArray(np.ascontiguousarray(a.point_to_value, dtype=np.uint32).view(np.int32), device)
            return cls(mesh, t, device)
        import torch
        dev = torch.device("cuda", device)
        t = {"faces": torch.from_numpy(mesh.faces.view(np.int32)).to(dev), "values": [], "maps": {}}
        for a in mesh.attributes:
            t["values"].append(torch.from_numpy(np.ascontiguousarray(a.values).view(np.int32)).to(dev))
            if a.point_to_value is not None and id(a.point_to_value) not in t["maps"]:
                t["maps"][id(a.point_to_value)] = torch.from_numpy(np.ascontiguousarray(a.point_to_value, dtype=np.uint32).view(np.int32)).to(dev)
        torch.cuda.synchronize(dev)
        return cls(mesh, t, device)

    def _c(self):
        m = self.mesh._c()
        m.faces = self._t["faces"].data_ptr()
        arr = m._keep
        for i, a in enumerate(self.mesh.attributes):
            arr[i].values = self._t["values"][i].data_ptr()
            if a.point_to_value is not None:
                arr[i].point_to_value = self._t["maps"][id(a.point_to_value)].data_ptr()
        return m


def encode_mesh_device(dmesh, cfg=None):
    """dmi_encode_mesh_device: whole .drc of a mesh resident in HBM."""
    L = load_library()
    cfg = cfg or Config(device=dmesh.device)
    m, c, out = dmesh._c(), cfg._c(), _Buffer()
    _check(L.dmi_encode_mesh_device(C.byref(m), C.byref(c), C.byref(out)))
    return _take(out)


def encode_mesh_device_raw(dmesh, cfg=None, cmesh=None):
    """encode_mesh_device at the cost of the C-ABI call alone: the `.drc` stays in the library-owned buffer (an EncodedBatch of one item).
    `cmesh`: a `dmesh._c()` made ahead of time (the ctypes structs of the call)."""
    L = load_library()
    cfg = cfg or Config(device=dmesh.device)
    m, c = cmesh if cmesh is not None else dmesh._c(), cfg._c()
    outs = (_Buffer * 1)()
    _check(L.dmi_encode_mesh_device(C.byref(m), C.byref(c), outs))
    return EncodedBatch(outs, 1)


def encode_connectivity(mesh):
    return Connectivity(mesh)


CONN_BAD_INDEX, CONN_DEGENERATE, CONN_NONMANIFOLD_EDGE, CONN_MULTI_FAN, CONN_UNUSED_VERTEX, CONN_HAS_BOUNDARY = 1, 2, 4, 8, 16, 32


def device_corner_table(mesh, cfg=None):
    """dmi_device_corner_table: the universal corner table of one mesh built by the device kernels (dmi_conn.hip), read back.
    → dict(num_vertices, opposite, left_most_corner, on_boundary, flags); num_vertices == 0 with flags != 0: the host builder's case."""
    L = load_library()
    cfg = cfg or Config.default()
    m, c = mesh._c(), cfg._c()
    nf = len(mesh.faces)
    cap = int(mesh.attributes[0].values.shape[0])
    opp = np.zeros(3 * nf, np.uint32)
    lmc = np.zeros(max(cap, 1), np.uint32)
    onb = np.zeros(max(cap, 1), np.uint8)
    nv, flags = C.c_uint32(0), C.c_uint32(0)
    _check(L.dmi_device_corner_table(C.byref(m), C.byref(c), opp.ctypes.data, lmc.ctypes.data, onb.ctypes.data, C.byref(nv), C.byref(flags)))
    return dict(num_vertices=nv.value, opposite=opp, left_most_corner=lmc[:nv.value], on_boundary=onb[:nv.value], flags=flags.value)


def device_attribute_table(mesh, att_index, cfg=None):
    """dmi_device_attribute_table: the corner table of attribute `att_index` of one mesh built by the device kernels (k_att_*), read back.
    → dict(num_vertices, interior_seams, seam_edge, corner_to_vertex, opposite, left_most_corner, flags); num_vertices == 0: flagged mesh."""
    L = load_library()
    cfg = cfg or Config.default()
    m, c = mesh._c(), cfg._c()
    nc = 3 * len(mesh.faces)
    seam, c2v, opp, lmc = np.zeros(nc, np.uint8), np.zeros(nc, np.uint32), np.zeros(nc, np.uint32), np.zeros(max(nc, 1), np.uint32)
    nv, interior, flags = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    _check(L.dmi_device_attribute_table(C.byref(m), C.byref(c), att_index, seam.ctypes.data, c2v.ctypes.data, opp.ctypes.data, lmc.ctypes.data, C.byref(nv), C.byref(interior), C.byref(flags)))
    return dict(num_vertices=nv.value, interior_seams=bool(interior.value), seam_edge=seam, corner_to_vertex=c2v, opposite=opp, left_most_corner=lmc[:nv.value], flags=flags.value)


def _decoded_attributes(atts, n):
    res = []
    for i in range(n):
        a = atts[i]
        raw = C.string_at(a.values, a.num_points * a.num_components * 4) if a.num_points else b""
        vals = np.frombuffer(raw, dtype=np.uint32 if a.portabilization == 1 else np.float32).reshape(a.num_points, a.num_components).copy()
        res.append(dict(att_type=a.att_type, num_components=a.num_components, unique_id=a.unique_id, domain=a.domain, scheme=a.scheme, transform=a.transform,
                        portabilization=a.portabilization, bits=a.bits, values=vals))
    return res


def decode_mesh(drc, cfg=None):
    """dmi_decode_mesh: a whole `.drc` from its bytes alone → dict(faces [F, 3] point indices in decode order, num_points,
    attributes [dict(att_type, …, values [num_points, num_components])])."""
    L = load_library()
    cfg = cfg or Config.default()
    b = np.frombuffer(bytes(drc), dtype=np.uint8)
    out, c = _DecodedMesh(), cfg._c()
    _check(L.dmi_decode_mesh(b.ctypes.data, len(b), C.byref(c), C.byref(out)))
    try:
        faces = np.ctypeslib.as_array(out.faces, shape=(3 * out.num_faces,)).copy().reshape(-1, 3) if out.num_faces else np.zeros((0, 3), np.uint32)
        return dict(faces=faces, num_points=out.num_points, attributes=_decoded_attributes(out.attributes, out.num_attributes))
    finally:
        L.dmi_decoded_mesh_free(C.byref(out))


def last_decode_timings():
    """dmi_last_decode_timings: host wall clock of this thread's last decode call by stage (ms)."""
    L = load_library()
    t = _DecodeTimings()
    _check(L.dmi_last_decode_timings(C.byref(t)))
    return {k: getattr(t, k) for k, _ in _DecodeTimings._fields_}


def decode_connectivity(header_and_connectivity):
    """dmi_decode_connectivity (host only): → dict(tables [dict like Connectivity.table, sequence empty], seeds, consumed)."""
    L = load_library()
    b = np.frombuffer(bytes(header_and_connectivity), dtype=np.uint8)
    conn, used = _Conn(), C.c_size_t(0)
    _check(L.dmi_decode_connectivity(b.ctypes.data, len(b), C.byref(conn), C.byref(used)))
    try:
        def arr(ptr, n):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_uint32)), shape=(n,)).copy() if ptr and n else np.zeros(0, np.uint32)
        tables = []
        for i in range(conn.num_tables):
            t = conn.tables[i]
            nc = 3 * t.num_faces
            tables.append(dict(num_faces=t.num_faces, num_vertices=t.num_vertices, corner_to_point=arr(t.corner_to_point, nc), corner_to_vertex=arr(t.corner_to_vertex, nc),
                               opposite=arr(t.opposite, nc), left_most_corner=arr(t.left_most_corner, t.num_vertices), sequence=np.zeros(0, np.uint32)))
        return dict(tables=tables, seeds=arr(conn.seeds, conn.num_seeds), consumed=used.value)
    finally:
        L.dmi_decoded_conn_free(C.byref(conn))


def shard_meshes(meshes, n_devices):
    """dmi_shard_meshes: device index per mesh, dealt by triangle count (LPT)."""
    L = load_library()
    n = len(meshes)
    arr = (_Mesh * max(n, 1))()
    keep = []
    for i, m in enumerate(meshes):
        cm = m._c()
        keep.append(cm)
        arr[i] = cm
    out = np.zeros(max(n, 1), np.int32)
    _check(L.dmi_shard_meshes(arr, n, n_devices, out.ctypes.data))
    return [int(x) for x in out[:n]]


def meshes_prepare_devices(meshes, device_of_mesh, cfg=None):
    """dmi_meshes_prepare_devices: mesh j prepared on HIP device device_of_mesh[j] (one process, several GPUs)."""
    L = load_library()
    cfg = cfg or Config.default()
    n = len(meshes)
    if n == 0:
        return []
    arr = (_Mesh * n)()
    keep = []
    for i, m in enumerate(meshes):
        cm = m._c()
        keep.append(cm)
        arr[i] = cm
    dev = np.ascontiguousarray(device_of_mesh, dtype=np.int32)
    assert len(dev) == n
    c = cfg._c()
    heads = (_Buffer * n)()
    handles = (C.c_void_p * n)()
    _check(L.dmi_meshes_prepare_devices(arr, n, C.byref(c), dev.ctypes.data, heads, handles))
    return [Job(handles[i], _take(heads[i])) for i in range(n)]


def jobs_encode_devices(jobs):
    """dmi_jobs_encode_devices: one dmi_jobs_encode per device, all devices concurrently; sections in job order."""
    L = load_library()
    n = len(jobs)
    handles = (C.c_void_p * n)(*[j._h for j in jobs])
    outs = (_Buffer * n)()
    _check(L.dmi_jobs_encode_devices(handles, n, outs))
    return [_take(outs[i]) for i in range(n)]


def host_rans_stream(freq, precision, symbols):
    """The hybrid form's host-core rANS coder on its own (no device): `symbols` coded last to first with the normalised
    frequencies `freq` (sum 2^precision); returns the stream bytes incl. the tagged final state."""
    L = load_library()
    f = np.ascontiguousarray(freq, dtype=np.uint32)
    s = np.ascontiguousarray(symbols, dtype=np.uint32)
    out = _Buffer()
    _check(L.dmi_host_rans_stream(f.ctypes.data, len(f), precision, s.ctypes.data, len(s), C.byref(out)))
    return _take(out)


def host_rabs_stream(zero_prob, bits):
    """The hybrid form's host-core rABS coder on its own: `bits` coded first to last with P(0) = zero_prob / 256."""
    L = load_library()
    b = np.ascontiguousarray(bits, dtype=np.uint8)
    out = _Buffer()
    _check(L.dmi_host_rabs_stream(zero_prob, b.ctypes.data, len(b), C.byref(out)))
    return _take(out)


def tile_sort_slots(sequence_to_point, tile_entries, block_entries=16384, cfg=None):
    """dmi_tile_sort_slots: → (slot_point, slot_entry) of the tile-sorted quantize gather."""
    L = load_library()
    cfg = cfg or Config.default()
    s2p = np.ascontiguousarray(sequence_to_point, dtype=np.uint32)
    sp, se = np.zeros(len(s2p), np.uint32), np.zeros(len(s2p), np.uint32)
    c = cfg._c()
    _check(L.dmi_tile_sort_slots(s2p.ctypes.data, len(s2p), tile_entries, block_entries, C.byref(c), sp.ctypes.data, se.ctypes.data))
    return sp, se


def host_rabs_constant_stream(zero_prob, bit, n):
    """n copies of `bit` through the rABS coder by its period (dmi_host_rabs_constant_stream): the bytes host_rabs_stream gives for them."""
    L = load_library()
    out = _Buffer()
    _check(L.dmi_host_rabs_constant_stream(C.c_uint8(zero_prob), C.c_uint32(bit), C.c_uint64(n), C.byref(out)))
    return _take(out)


def decode_attributes(section, tables, num_points, seeds=None, cfg=None):
    """dmi_decode_attributes: the attribute section `section` read back against the corner tables of the connectivity stage (the
    arrays encode_attributes takes).  Returns [dict(att_type, num_components, unique_id, scheme, transform, portabilization, bits,
    values [num_points, num_components] float32 — uint32 for a ToBits attribute)]."""
    L = load_library()
    cfg = cfg or Config.default()
    b = np.frombuffer(section, dtype=np.uint8)
    tabs, keep = _tables_c(tables)
    sd = None if seeds is None else np.ascontiguousarray(seeds, dtype=np.uint32)
    out, c = _Decoded(), cfg._c()
    _check(L.dmi_decode_attributes(b.ctypes.data, len(b), tabs, len(tables), None if sd is None else sd.ctypes.data, 0 if sd is None else len(sd), num_points, C.byref(c), C.byref(out)))
    try:
        return _decoded_attributes(out.attributes, out.num_attributes)
    finally:
        L.dmi_decoded_free(C.byref(out))
