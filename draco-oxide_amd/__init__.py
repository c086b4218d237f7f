"""draco-oxide_amd — MI355X-native attribute-encoding hot path of draco-oxide.

Python here is plumbing only: a ctypes binding of libdraco_mi.so (C ABI in include/draco_mi.h)
whose names mirror the reference crate (`encode::encode(mesh, &mut buf, Config::default())`,
`Mesh`, `Attribute`, `MeshBuilder`).  All arithmetic runs in the HIP kernels under csrc/.
The package directory name contains a hyphen, so import it through the `draco_oxide_amd` shim at
the repository root.
"""
from .binding import (  # noqa: F401
    ATT_POSITION, ATT_NORMAL, ATT_COLOR, ATT_TEXCOORD, ATT_CUSTOM, DOMAIN_POSITION, DOMAIN_CORNER,
    F32, U32, I32, FLAG_TIMINGS, POS_SCHEME_DELTA,
    Attribute, Mesh, MeshBuilder, Config, DracoMiError, Job, Connectivity,
    encode, encode_mesh, encode_attributes, encode_attributes_batch, encode_connectivity, mesh_prepare, meshes_prepare, jobs_encode, jobs_encode_raw, EncodedBatch, device_count, release_cached_memory, library_path, load_library,
    host_rans_stream, host_rabs_stream, host_rabs_constant_stream, tile_sort_slots, decode_attributes, decode_mesh, decode_connectivity, last_decode_timings, shard_meshes, meshes_prepare_devices, jobs_encode_devices, device_corner_table, DeviceMesh, encode_mesh_device, encode_mesh_device_raw, last_call_timings, init,
    RawMesh, BuiltBatch, Transcoder, meshes_build, built_meshes_prepare, last_build_timings, device_attribute_table,
    configure_process, debug_from_env,
)
from . import binding, gltf, synth  # noqa: E402,F401
