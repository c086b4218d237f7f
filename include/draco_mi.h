/* draco_mi.h — C ABI of libdraco_mi.so: MI355X-native drop-in for draco-oxide's attribute-encoding
 * hot path (quantization → prediction → prediction transform → rANS symbol coding).
 *
 * Every entry point names the reference interface it replaces (paths relative to
 * reearth/draco-oxide `draco-oxide/src/`).  The reference has no FFI; the seam a maintainer would
 * bind is the internal call `attribute::encode_attributes(attributes, writer, conn_out, &cfg)`
 * (encode/mod.rs:90), fed by `connectivity::encode_connectivity` (encode/mod.rs:86).  INTEGRATION.md
 * shows the Rust shim.
 *
 * Conventions: plain pointers and sizes, no C++/torch types; all index arrays are uint32;
 * DMI_NONE (0xFFFFFFFF) means "no corner"; every function returns a dmi_status (0 = ok) and
 * never panics/aborts across the boundary — each reference panic path is an error code.
 * Output buffers are library-owned: release them with dmi_free().
 */
#ifndef DRACO_MI_H
#define DRACO_MI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DMI_NONE 0xFFFFFFFFu

typedef enum dmi_status {
  DMI_OK = 0,
  DMI_ERR_INVALID_ARGUMENT = 1,
  DMI_ERR_UNSUPPORTED_DATA_TYPE = 2,      /* attribute_encoder.rs:31  Err::UnsupportedDataType */
  DMI_ERR_UNSUPPORTED_NUM_COMPONENTS = 3, /* attribute_encoder.rs:33  Err::UnsupportedNumComponents */
  DMI_ERR_PARENT_NOT_ENCODED = 4,         /* encode/attribute/mod.rs:65 unwrap() panic */
  DMI_ERR_BAD_PARENT = 5,                 /* mesh_normal_prediction.rs:57-61 asserts */
  DMI_ERR_ZERO_NORMAL = 6,                /* prediction_transform/geom.rs:45 assert */
  DMI_ERR_ENTROPY = 7,                    /* encode/entropy/rans.rs:259-269 Err (StateTooLarge, ...) */
  DMI_ERR_ALPHABET_TOO_LARGE = 8,
  DMI_ERR_NO_DEVICE = 9,                  /* no HIP device / HIP runtime failure: the library has NO CPU fallback */
  DMI_ERR_HIP = 10,
  DMI_ERR_OUT_OF_MEMORY = 11,
  DMI_ERR_CONNECTIVITY = 12,              /* encode/connectivity/edgebreaker.rs Err / corner_table panics */
  DMI_ERR_UNUSED_VERTICES = 13,           /* core/corner_table/mod.rs:105-108 panic */
  DMI_ERR_IO = 14
} dmi_status;

/* Wire ids — identical to the reference's writers. */
enum { DMI_ATT_POSITION = 0, DMI_ATT_NORMAL = 1, DMI_ATT_COLOR = 2, DMI_ATT_TEXCOORD = 3, DMI_ATT_CUSTOM = 4,
       DMI_ATT_TANGENT = 5, DMI_ATT_MATERIAL = 6, DMI_ATT_JOINT = 7, DMI_ATT_WEIGHT = 8 };   /* core/attribute/mod.rs:648-661 */
enum { DMI_DOMAIN_POSITION = 0, DMI_DOMAIN_CORNER = 1 };                                      /* core/attribute/mod.rs:705-710 */
enum { DMI_U8 = 1, DMI_I8 = 2, DMI_U16 = 3, DMI_I16 = 4, DMI_U32 = 5, DMI_I32 = 6, DMI_U64 = 7, DMI_I64 = 8,
       DMI_F32 = 9, DMI_F64 = 10 };                                                           /* core/attribute/mod.rs:568-582 */

/* One mesh attribute as the reference's `Attribute` holds it (core/attribute/mod.rs:26-49):
 * unique values (AoS) + optional point→value map. */
typedef struct dmi_attribute {
  const void* values;             /* num_unique rows of num_components components, host memory */
  uint32_t num_unique;
  uint8_t component_type;         /* DMI_F32 (quantized paths) or a 4-byte integer type (DMI_ATT_CUSTOM / ToBits) */
  uint8_t num_components;         /* 1..4 */
  uint8_t att_type;               /* DMI_ATT_* */
  uint8_t domain;                 /* DMI_DOMAIN_* */
  uint32_t unique_id;             /* AttributeId, written as the "unique id" byte */
  int32_t parent_index;           /* index in the attribute array of parents[0], -1 = none (Attribute::get_parents) */
  const uint32_t* point_to_value; /* num_points entries, or NULL = identity (point_to_att_val_map: None) */
  uint32_t num_points;            /* Attribute::len() */
} dmi_attribute;

/* Flat view of the corner table the reference hands to the attribute encoder
 * (core/corner_table/mod.rs:8-52 GenericCornerTable; universal table for attribute 0, the
 * RefAttributeCornerTable of attribute i-1 otherwise: all_inclusive_corner_table.rs:31-45). */
typedef struct dmi_corner_table {
  uint32_t num_faces;
  uint32_t num_vertices;
  const uint32_t* corner_to_point;   /* 3F  point_idx()   = mesh faces            corner_table/mod.rs:484-487 */
  const uint32_t* corner_to_vertex;  /* 3F  vertex_idx()                           :443-459 / attribute_corner_table.rs:144 */
  const uint32_t* opposite;          /* 3F  opposite(), DMI_NONE across boundaries and (attribute tables) seams */
  const uint32_t* left_most_corner;  /* num_vertices                                                          */
  const uint32_t* sequence;          /* optional: Traverser::compute_seqeunce() output (shared/attribute/sequence.rs:48);
                                        NULL → computed by the library from `seeds` */
  uint32_t sequence_len;
} dmi_corner_table;

/* Path-selecting and tuning switches — the typed form (round 6) of what were ≈ 50 environment variables read all over the library.  NONE changes the
 * bitstream: each selects another implementation of the same stage (the tests run every form against the oracle) or sizes a heuristic.  Hung off
 * dmi_config: read once per call, kept per job — two concurrent calls may differ.  The library itself never reads the environment for these: a caller
 * that wants `DMI_NO_FUSED=1 ./app` behaviour fills the struct from its own environment (draco-oxide_amd/binding.py does, per call, for tests / bench).
 * The reference's nested Config (encode/mod.rs:22-42, attribute_encoder.rs:113-136, portabilization/mod.rs:111-143) carries its own choices the same
 * way: as typed fields of the value handed to encode(). */
typedef struct dmi_debug {
  uint64_t flags;             /* DMI_DBG_* */
  uint32_t host_threads;      /* host threads a call may keep busy (0 = the machine's, no more than the cgroup quota) */
  int32_t  tile_sort;         /* quantize gather of resident jobs: 0 default (sorted tiles for large resident jobs), -1 off, > 0 tile entries */
  uint32_t tile_sort_min, tile_sort_local;   /* 0 = defaults: sequence length the sort starts at / entries one workgroup sorts in LDS */
  uint32_t seq_big_entries;   /* 0 = default: sequence length from which k_seq_quantize_big runs */
  uint8_t  relabel;           /* coding-order relabelling: 0 by size, 1 device, 2 host */
  uint8_t  chains;            /* serial coders of a single job: 0 by the longest stream, 1 device walker, 2 host cores */
  uint8_t  prepare_threads;   /* transcoder: threads of the prepare step (0 = default 2) */
  uint8_t  pad1;
  uint32_t fused_grid, fused_lds, chain_grid;   /* tuning aids of the sweep / chain launches (0 = defaults) */
  uint32_t batch_threads, split;                /* batch encode: worker threads, sub-batches (0 = defaults) */
  uint32_t shadow_min_faces;                    /* host walks: faces from which the shadow prefetch runs (0 = default) */
  uint64_t prep_group_faces, batch_min_faces;   /* batch prepare: faces per device group, smallest mesh that takes the device tables (0 = defaults) */
  uint32_t stage_primitives, stage_ramp;        /* transcoder: a stage is dispatched at this many primitives even below its triangle count (0 = no cap);
                                                   stage_ramp n > 0: the first stage takes 1/n of a stage's triangles and the following ones double up to a whole stage */
} dmi_debug;
#define DMI_DBG_NO_FUSED          (1ull << 0)    /* per-attribute predictor kernels instead of the fused sweep */
#define DMI_DBG_NO_PACKED         (1ull << 1)    /* int32 quantized values instead of the sweep's packed layouts */
#define DMI_DBG_NO_SYM16          (1ull << 2)    /* uint32 symbols */
#define DMI_DBG_NO_EARLY          (1ull << 3)    /* no early stage in whole-mesh calls */
#define DMI_DBG_NO_PLAIN_ORDER    (1ull << 4)    /* one-shot jobs re-order their faces like resident ones */
#define DMI_DBG_HOST_TABLES       (1ull << 5)    /* frequency tables normalised on the host */
#define DMI_DBG_HOST_CONNECTIVITY (1ull << 6)    /* corner tables by the host builder */
#define DMI_DBG_HOST_ATT_TABLES   (1ull << 7)    /* attribute corner tables by the host builder */
#define DMI_DBG_HOST_BUILD        (1ull << 8)    /* MeshBuilder::build by the host builder */
#define DMI_DBG_NO_IN_PLACE       (1ull << 9)    /* pack + copy even what could go up where it lies */
#define DMI_DBG_NO_POOL           (1ull << 10)   /* a hipMalloc per job buffer */
#define DMI_DBG_POISON            (1ull << 11)   /* uncleared chunks filled with 0xA5 */
#define DMI_DBG_ZERO_CHUNKS       (1ull << 12)   /* job chunks cleared as a whole */
#define DMI_DBG_NO_QUAD           (1ull << 13)   /* the host walks read 3·face + k ids */
#define DMI_DBG_TEST_QUAD         (1ull << 14)   /* (tests, host only) 4·face + k ids on host-built tables */
#define DMI_DBG_NO_CLOSED         (1ull << 15)   /* closed meshes run the general walk loops */
#define DMI_DBG_NO_SHADOW         (1ull << 16)   /* Edgebreaker walk without the shadow prefetch */
#define DMI_DBG_NO_SEQ_SHADOW     (1ull << 17)   /* sequencer without it */
#define DMI_DBG_NO_SEAM_MASKS     (1ull << 18)   /* seam streams from per-corner flags instead of per-face masks */
#define DMI_DBG_NO_DEFER_SEAMS    (1ull << 19)   /* meshes with attribute seams are prepared one by one */
#define DMI_DBG_NO_BATCHED_PHASES (1ull << 20)   /* a batch encodes job by job */
#define DMI_DBG_FUSED_WINDOWS     (1ull << 21)   /* the sweep with LDS-staged neighbourhoods (recorded experiment) */
#define DMI_DBG_CHAIN_DENSE       (1ull << 22)   /* dense chain launches only */
#define DMI_DBG_SERIAL_TABLES     (1ull << 23)   /* host corner tables on one thread */
#define DMI_DBG_PARALLEL_TABLES   (1ull << 24)   /* … on all threads whatever the size */
#define DMI_DBG_FILE_ORDER        (1ull << 25)   /* dmi_transcode_assets takes the files in the caller's order */
#define DMI_DBG_TRACE             (1ull << 26)   /* stage lines on stderr */
#define DMI_DBG_TRACE_STAGES      (1ull << 27)   /* the transcoder's stage lines only */
#define DMI_DBG_TRACE_TABLES      (1ull << 28)
#define DMI_DBG_BUILD_TRACE       (1ull << 29)
#define DMI_DBG_NO_SEQ_STREAM      (1ull << 30)   /* whole-mesh calls upload the sequence after the sequencer, in one copy */
#define DMI_DBG_NO_STREAM_COPY   (1ull << 33)   /* the build's pack of pageable accessors into staging by plain memcpy (default: non-temporal stores) */
#define DMI_DBG_SPIN_WAITS       (1ull << 32)   /* the batch path's long device waits spin like every other wait (default: blocking events — a waiting thread leaves its core to the walks) */
#define DMI_DBG_SMALL_HEAD       (1ull << 31)   /* dmi_transcode_assets: a device's smallest files (1/32 of its bytes) go first, the rest largest first */
typedef struct dmi_config {
  uint8_t pos_bits;      /* 0 → 11   portabilization/mod.rs:118-121 */
  uint8_t uv_bits;       /* 0 → 10   :131-134 */
  uint8_t generic_bits;  /* 0 → 11 */
  uint8_t pos_scheme;    /* 0 → reference default (1 = MeshParallelogram + WrappedDifference);
                            0xD0 → DeltaPrediction + Difference (BASELINE config 2 variant, not selectable in the reference) */
  int32_t device;        /* HIP device ordinal */
  void* stream;          /* hipStream_t to launch on, NULL = a stream owned by the job */
  uint32_t flags;        /* DMI_FLAG_* */
  const dmi_debug* debug;   /* NULL = the process defaults (all zeros unless dmi_set_default_debug changed them) */
} dmi_config;
#define DMI_FLAG_TIMINGS 1u   /* record per-stage hipEvent timings (dmi_job_timings) */
/* the switches of calls that carry none (debug == NULL, or no dmi_config at all): copied; NULL = back to all zeros */
void dmi_set_default_debug(const dmi_debug* d);
/* Process-wide behaviour the library does NOT take upon itself by default (round 6: a drop-in must not rearrange its host process): */
#define DMI_PROCESS_HUGE_PAGE_NEW 1u   /* large `operator new` blocks of the library's own containers on 2 MiB-aligned, huge-page-advised memory (host_pool.cpp) */
#define DMI_PROCESS_NUMA_PIN      2u   /* whole-mesh calls restrict the calling thread to the CPUs of the GPU's memory node for the duration of the call */
#define DMI_PROCESS_NO_THP        4u   /* never advise transparent huge pages */
typedef struct dmi_process_options { uint32_t flags; uint32_t host_cache_mb, device_cache_mb, decode_budget_mb; /* 0 = defaults (4096 / 8192 / 16384) */ } dmi_process_options;
int dmi_configure_process(const dmi_process_options* o);   /* any time; affects calls made afterwards */

typedef struct dmi_buffer { uint8_t* data; size_t len; size_t cap; } dmi_buffer;

/* Per-stage device times of the last dmi_job_encode (milliseconds, hipEvent-measured on the job's stream). */
typedef struct dmi_timings {
  float quantize_ms;     /* min/max + quantize kernels, all attributes */
  float predict_ms;      /* rank/gather + predict+transform kernels (the HBM-roofline pass) */
  float histogram_ms;
  float table_ms;        /* table stage + record prep: k_tables and the prep kernels on the device (host-table form: D2H histogram,
                            host normalisation, H2D tables — host-inclusive wall time) */
  float rans_ms;         /* rANS + rABS coders: the chain kernel (device form), or read-back + host-core chains (hybrid form, wall clock) */
  float total_ms;        /* first launch → last byte on host */
  uint64_t predict_bytes;    /* algorithmic bytes of the quantize+predict pass (SURVEY §8d formula) */
  uint64_t symbols;          /* rANS symbols coded */
  uint32_t num_streams;      /* rANS + rABS chains */
  uint32_t host_chains;      /* 1 = the streams of this encode were coded on host cores (hybrid form, long single meshes), 0 = on the device */
  float longest_stream_ms;   /* hybrid form: the coder of the longest stream alone (its symbols already on the host); device form: 0 */
  float readback_wait_ms;    /* hybrid form: time the longest stream's host thread waited for its symbols / tables to arrive */
  /* whole-mesh calls only (dmi_last_call_timings): host wall clock of the stages around the encode */
  float mesh_readback_ms;    /* dmi_encode_mesh_device: faces + maps read back for the host walks */
  float tables_ms;           /* universal corner table: device kernels + read-back (or the host builder) */
  float connectivity_ms;     /* the whole connectivity stage, tables_ms included: Edgebreaker traversal, connectivity bytes, sequencers */
  float job_create_ms;       /* coding-order relabelling, uploads, fan rows, buffers */
  float call_ms;             /* the whole call */
  uint32_t texcoord_fixups;  /* texture-coordinate entries of the fused sweep whose operands were outside its exact f64 tier: predicted by the general i64 form (k_texcoord_fixup) */
  float job_create_device_ms;   /* whole-mesh calls: device span of job creation (first launch → last, hipEvents on the job's stream): coding-order
                                   relabelling, map compositions, fan rows, buffer clears — what the quantize+predict pass presupposes per call */
  float early_ms;               /* dmi_encode_mesh_device: value ranges + value-order quantization issued on a side stream BEFORE the host's serial
                                   walks (hipEvents on that stream); 0 when the call had no early stage.  quantize_ms then covers the coding-order
                                   gather of the packed values only */
} dmi_timings;
/* Timings of the last dmi_encode_mesh / dmi_encode_mesh_device / dmi_encode_attributes call of the calling thread (per-stage device
 * times when that call's dmi_config carried DMI_FLAG_TIMINGS). */
int dmi_last_call_timings(dmi_timings* t);

/* --- Drop-in for attribute::encode_attributes (encode/attribute/mod.rs:13-93) ------------------
 * atts[i] is encoded against tables[i]; `seeds` = Output::corners_of_edgebreaker
 * (encode/connectivity/edgebreaker.rs:98-101,523-529).  Appends nothing to caller memory: the
 * attribute section bytes are returned in `out`. Host pointers in, host bytes out. */
int dmi_encode_attributes(const dmi_attribute* atts, const dmi_corner_table* tables, uint32_t n_atts,
                          const uint32_t* seeds, uint32_t n_seeds, const dmi_config* cfg, dmi_buffer* out);

/* --- Resident form of the same call: upload once, encode many times (bench / pipelines). -------
 * dmi_job_create copies every input to HBM (and computes missing sequences on the host);
 * dmi_job_encode runs quantize → predict → transform → histogram → rANS entirely from HBM. */
typedef struct dmi_job dmi_job;
int dmi_job_create(const dmi_attribute* atts, const dmi_corner_table* tables, uint32_t n_atts,
                   const uint32_t* seeds, uint32_t n_seeds, const dmi_config* cfg, dmi_job** job);
int dmi_job_encode(dmi_job* job, dmi_buffer* out);
int dmi_job_timings(const dmi_job* job, dmi_timings* t);
void dmi_job_destroy(dmi_job* job);

/* --- Batches of independent meshes (the glTF transcoder calls encode::encode once per primitive:
 * io/gltf/encode.rs:932-955,1827-1842).  All jobs must live on one device.  The phases, the table stage and the record prep
 * of ALL jobs are planned together (one upload, one launch per kernel), every rANS/rABS stream of every job runs in ONE
 * persistent launch, and the host waits once — for the packed read-back.  outs[j] receives job j's attribute section. */
int dmi_jobs_encode(dmi_job** jobs, uint32_t n_jobs, dmi_buffer* outs);
typedef struct dmi_batch_item {
  const dmi_attribute* atts;
  const dmi_corner_table* tables;
  uint32_t n_atts;
  const uint32_t* seeds;
  uint32_t n_seeds;
} dmi_batch_item;
/* dmi_encode_attributes for n meshes at once (host pointers in, n attribute sections out). */
int dmi_encode_attributes_batch(const dmi_batch_item* items, uint32_t n, const dmi_config* cfg, dmi_buffer* outs);

/* --- Drop-in for encode::encode(mesh, writer, Config::default()) (encode/mod.rs:59-97) ----------
 * Host: header, Edgebreaker connectivity, corner tables, sequencer.  Device: attribute section.
 * `faces` are point indices (Mesh::faces); atts[0] must be the Position attribute
 * (MeshBuilder::get_sorted_attributes, core/mesh/builder.rs:115-125). */
typedef struct dmi_mesh {
  const uint32_t* faces;   /* 3 * num_faces point indices */
  uint32_t num_faces;
  const dmi_attribute* atts;
  uint32_t num_atts;
} dmi_mesh;
int dmi_encode_mesh(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* out);

/* The same call for a mesh that already lives in HBM: `mesh` is a host struct whose `faces`, `atts[i].values` and
 * `atts[i].point_to_value` are DEVICE pointers on cfg->device.  Faces and maps are read back once (the Edgebreaker traversal and the
 * attribute sequencer are serial host walks); the attribute values never leave the device.  bench.py's `value` times this call. */
int dmi_encode_mesh_device(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* out);

/* Host stages of dmi_encode_mesh split out, so a caller (bench.py, a batch driver) can keep the
 * serial graph walks outside a timed/pipelined region: returns the connectivity bytes (header
 * included) and a resident job for the attribute section. */
int dmi_mesh_prepare(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** job);

/* dmi_mesh_prepare for n independent meshes (the glTF transcoder's primitives, io/gltf/encode.rs:932-955): the host
 * graph walks and uploads of different meshes run concurrently on a pool of host threads.  All-or-nothing: on error no
 * job or buffer is left allocated. */
int dmi_meshes_prepare(const dmi_mesh* meshes, uint32_t n, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** jobs);

/* --- One process, several GPUs (a single-process caller — the Rust crate — has no torch.distributed): the same batch calls with a
 * device per mesh / job.  dmi_shard_meshes deals n meshes over n_devices by triangle count (greedy longest-processing-time, the
 * partition of the one-process-per-GPU form); dmi_meshes_prepare_devices prepares mesh j on HIP device device_of_mesh[j]
 * (dmi_config.device is ignored, dmi_config.stream must be null); dmi_jobs_encode_devices groups the jobs by their device and runs
 * one dmi_jobs_encode per device concurrently.  outs[j] / jobs[j] / header_and_connectivity[j] stay in mesh order.  All or nothing. */
int dmi_shard_meshes(const dmi_mesh* meshes, uint32_t n, uint32_t n_devices, int32_t* device_of_mesh);
int dmi_meshes_prepare_devices(const dmi_mesh* meshes, uint32_t n, const dmi_config* cfg, const int32_t* device_of_mesh, dmi_buffer* header_and_connectivity, dmi_job** jobs);
int dmi_jobs_encode_devices(dmi_job** jobs, uint32_t n_jobs, dmi_buffer* outs);

/* Host connectivity only (no GPU needed): header + connectivity bytes, plus the flat tables that
 * dmi_encode_attributes consumes.  Tables are library-owned and freed by dmi_conn_free. */
typedef struct dmi_conn {
  uint32_t num_tables;             /* 1 + number of non-position attributes */
  dmi_corner_table* tables;        /* tables[0] = universal, tables[j] = attribute table j-1; sequences filled in */
  uint32_t* seeds;
  uint32_t num_seeds;
  void* owner;
} dmi_conn;
int dmi_encode_connectivity(const dmi_mesh* mesh, dmi_buffer* header_and_connectivity, dmi_conn* conn);
void dmi_conn_free(dmi_conn* conn);

/* The order-free half of the connectivity stage on the device (core/corner_table/mod.rs:252-340 compute_table, :342-416
 * compute_left_most_corners, :36-38 is_on_boundary) for ONE mesh, read back: opposite[3F], left_most_corner[V] (nullable),
 * on_boundary[V] (nullable; sized for the Position attribute's value count).  dmi_mesh_prepare / dmi_meshes_prepare run the same
 * kernels internally (batched: one launch per kernel for all meshes).  *flags != 0 with *num_vertices == 0: the mesh has vertex-degenerate
 * faces, an edge with more than two faces or a vertex with several fans — the result then depends on the corner order and the
 * library takes the reference's serial walks on the host instead (dmi_encode_connectivity's tables); nothing was written. */
int dmi_device_corner_table(const dmi_mesh* mesh, const dmi_config* cfg, uint32_t* opposite, uint32_t* left_most_corner, uint8_t* on_boundary,
                            uint32_t* num_vertices, uint32_t* flags);

/* One attribute corner table (core/corner_table/attribute_corner_table.rs:16-137) of ONE mesh built by the device kernels, read back: seam
 * flags per corner (1 = the edge opposite the corner is a seam of attribute att_index or a boundary), attribute vertex per corner, opposite corner
 * (DMI_NONE across seams), left-most corner per attribute vertex (room for 3·num_faces entries).  dmi_meshes_prepare / dmi_built_meshes_prepare
 * run the same kernels for every (mesh, attribute) whose point → value map is not the position map entry for entry — batched, and the tables
 * stay on the device for the coding-order relabelling.  *num_vertices == 0 with *flags != 0: the universal table's kernels flagged the mesh
 * (dmi_device_corner_table) and the host builds its tables. */
int dmi_device_attribute_table(const dmi_mesh* mesh, const dmi_config* cfg, uint32_t att_index, uint8_t* seam_edge, uint32_t* corner_to_vertex, uint32_t* opposite,
                               uint32_t* left_most_corner, uint32_t* num_vertices, uint32_t* interior_seams, uint32_t* flags);

/* --- MeshBuilder::build (core/mesh/builder.rs:62-90) + Attribute::from's value dedup (core/attribute/mod.rs:394-452) ----
 * Host only.  Attributes in add order (AttributeId = index, builder.rs:31-39), one row per point; the result is the `Mesh` the
 * reference hands to encode::encode: unique values in first-occurrence order with point_to_value maps, Position swapped to
 * slot 0, identical points merged, degenerate faces and unreferenced points removed.  Arrays are owned by the library. */
typedef struct dmi_raw_attribute {
  const void* data;          /* num_points rows of num_components components */
  uint32_t num_points;
  uint8_t component_type, num_components, att_type, domain;
  uint32_t num_parents;
  const uint32_t* parents;   /* ids (add-order indices) of the parent attributes */
} dmi_raw_attribute;
typedef struct dmi_built_mesh { dmi_mesh mesh; void* owner; } dmi_built_mesh;
int dmi_mesh_build(const dmi_raw_attribute* atts, uint32_t n_atts, const uint32_t* faces, uint32_t num_faces, dmi_built_mesh* out);
void dmi_built_mesh_free(dmi_built_mesh* m);

/* --- Host memory the device may read in place (round 5) ----------------------------------------------------------------------------------
 * The reference reads an asset into a Vec<u8> (io/gltf/transcoder.rs:134-151 → read_scene_from_file / _from_buffer) and copies every accessor
 * out of it (io/gltf/decode.rs:2277-2309) before MeshBuilder sees a value.  Here the importer reads the file INTO memory from dmi_host_alloc —
 * page-locked blocks on huge pages, registered with the runtime once and recycled (dmi_host_free parks a block; dmi_release_cached_memory hands
 * the parked ones back) — and dmi_meshes_build / dmi_transcoder / dmi_transcode_assets copy the accessors' bytes up where they lie: the arrays'
 * ranges merged into spans (about one DMA per file), rows read with their byteStride, index arrays (u8 / u16 widened) gathered on the device.
 * No host pack, no staging copy.  Arrays in any other memory are packed into staging by host threads as before — same result either way.
 * (Page-locking the CALLER's own buffers per call was tried and withdrawn: see csrc/dmi_hostmem.cpp.) */
void* dmi_host_alloc(size_t bytes);       /* NULL on failure; contents unspecified */
void dmi_host_free(void* p);
int dmi_host_is_registered(const void* p, size_t bytes);   /* 1: [p, p + bytes) lies inside a block dmi_host_alloc handed out */

/* --- MeshBuilder::build for a BATCH of primitives, on the device ---------------------------------------------------------
 * What the glTF importer does once per triangle primitive (io/gltf/decode.rs:2328-2525: accessors → MeshBuilder::add_attribute →
 * build()) for n primitives in one call: the accessors' rows and the index arrays go up once (pinned staging), every step of
 * MeshBuilder::build — Attribute::from's value dedup with `==` classes (core/attribute/mod.rs:394-452), the merge of points that agree in
 * every attribute (core/mesh/builder.rs:194-279), degenerate faces (:77-79), unreferenced points (:129-189) — runs as ONE launch per
 * kernel for all primitives (dmi_build.hip: hash-based class search whose result does not depend on scheduling, prefix-sum ranks), and
 * the faces and point → value maps come back in one read-back (the host's serial walks need them).  The unique values stay in device
 * memory unless DMI_BUILD_HOST_VALUES is set.  out[j].mesh equals what dmi_mesh_build returns for primitive j (same value order, maps,
 * surviving points and faces, Position in slot 0) with atts[i].values == NULL when the values stayed on the device.  Primitives outside
 * the device form's class (attributes of different point counts, components that are not 4 bytes wide, more than 16 attributes, a face
 * index out of range, no surviving face) take the host builder inside the same call.  Free every out[j] with dmi_built_mesh_free. */
typedef struct dmi_raw_accessor {
  const void* data;          /* first element, host memory */
  uint32_t count;            /* elements = points */
  uint32_t byte_stride;      /* distance between elements, 0 = tightly packed (glTF bufferView.byteStride) */
  uint8_t component_type, num_components, att_type, domain;
  uint32_t num_parents;
  const uint32_t* parents;   /* ids (add-order indices) of the parent attributes */
} dmi_raw_accessor;
typedef struct dmi_raw_mesh {
  const dmi_raw_accessor* atts;   /* in add order: AttributeId = index (builder.rs:31-39) */
  uint32_t n_atts;
  const void* indices;            /* 3·num_faces point indices */
  uint8_t index_type;             /* DMI_U8 / DMI_U16 / DMI_U32 (glTF 5121 / 5123 / 5125) */
  uint32_t num_faces;
} dmi_raw_mesh;
#define DMI_BUILD_HOST_VALUES 1u   /* also read the unique values back: out[j].mesh is a complete host Mesh */
int dmi_meshes_build(const dmi_raw_mesh* raw, uint32_t n, const dmi_config* cfg, uint32_t flags, dmi_built_mesh* out);
/* Face and point counts of n built meshes (num_faces[j] == 0: the reference skips such a primitive, io/gltf/encode.rs:934-936), and the release of
 * all of them, each in ONE call. */
int dmi_built_meshes_info(const dmi_built_mesh* built, uint32_t n, uint32_t* num_faces, uint32_t* num_points);
void dmi_built_meshes_free(dmi_built_mesh* built, uint32_t n);
/* dmi_meshes_prepare for meshes that dmi_meshes_build left resident on cfg->device: nothing is packed or uploaded again — the connectivity
 * kernels read the built faces and maps where they are, the jobs copy their values device to device.  Same bytes as dmi_meshes_prepare on
 * the equivalent host meshes.  The built meshes may be freed as soon as this returns. */
int dmi_built_meshes_prepare(const dmi_built_mesh* built, uint32_t n, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** jobs);
/* --- The transcoder's per-primitive loop as ONE object (io/gltf/transcoder.rs:134-151, io/gltf/encode.rs:932-955,1827-1842) -------------------------
 * The importer pushes triangle primitives as it produces them (descriptors are copied; the accessor / index arrays they point to must stay
 * valid until the primitive is reported done); stages of ≈ stage_triangles triangles (0: a quarter of expected_triangles, within 3M … 12M;
 * the first stage a third of that) run dmi_meshes_build → dmi_built_meshes_prepare → dmi_jobs_encode on three library threads, stage k+2 / k+1 / k
 * side by side (two threads per step since round 5: stages may finish out of push order).  `done(user, first, count)` is called from a library thread — calls for different stages may come from two threads at once — when the primitives [first, first + count) (push order) are final:
 * dmi_transcoder_result then gives primitive i's header + connectivity bytes and attribute section — blob = the two back to back, what
 * dmi_encode_mesh writes for the built mesh — and its face / point counts for the placeholder accessors (num_faces == 0: no face left, the
 * reference leaves such a primitive alone, io/gltf/encode.rs:934-936; both buffers empty).  The buffers are the transcoder's until
 * dmi_transcoder_destroy.  dmi_transcoder_reserve(total primitives) is a hint; dmi_transcoder_finish flushes the last stage,
 * waits for everything and returns the first error of any stage (dmi_last_error() of the calling thread names it).  One device (cfg->device). */
typedef struct dmi_transcoder dmi_transcoder;
typedef void (*dmi_transcode_done_fn)(void* user, uint32_t first, uint32_t count);
dmi_transcoder* dmi_transcoder_create(const dmi_config* cfg, uint64_t expected_triangles, uint64_t stage_triangles, dmi_transcode_done_fn done, void* user);
int dmi_transcoder_reserve(dmi_transcoder* t, uint32_t n_primitives);
int dmi_transcoder_push(dmi_transcoder* t, const dmi_raw_mesh* prims, uint32_t n);
int dmi_transcoder_finish(dmi_transcoder* t);
int dmi_transcoder_result(dmi_transcoder* t, uint32_t i, dmi_buffer* header_and_connectivity, dmi_buffer* section, uint32_t* num_faces, uint32_t* num_points);
int dmi_transcoder_timings(dmi_transcoder* t, double* build_ms, double* prepare_ms, double* encode_ms);   /* time inside the three calls, summed over the stages */
/* how the primitives built so far were built: by the device kernels, by the host builder inside the same call (outside the device form's class, or
 * flagged by the kernels), and — of the former — copied up where they lay (dmi_host_alloc memory) */
int dmi_transcoder_counts(dmi_transcoder* t, uint64_t* device_built, uint64_t* host_built, uint64_t* in_place);
uint32_t dmi_transcoder_stages(dmi_transcoder* t);   /* pipeline stages dispatched so far */
void dmi_transcoder_destroy(dmi_transcoder* t);

/* --- A list of glTF assets in, their Draco-compressed GLBs out (round 5; io/gltf/transcoder.rs:134-151 per file: read_scene → compress_scene →
 * write_scene) --------------------------------------------------------------------------------------------------------------------------------
 * An asset is a GLB container (`glb`), or a glTF JSON document with its buffers already resolved by the caller (files, data URIs).  Every triangle
 * primitive with a POSITION becomes one Mesh exactly as the reference's importer builds it (io/gltf/decode.rs:2328-2525: attributes in sorted
 * semantic order, raw little-endian f32 rows with the view's stride, NORMAL / TEXCOORD_0 as Corner attributes whose parent is the position,
 * `_FEATURE_ID_n` as Custom u32 attributes; a primitive for which the reference hands out a wrong parent id, or one that is Draco-compressed
 * already, fails the call), is coded as dmi_encode_mesh codes it, and its file is rewritten as io/gltf/encode.rs:932-1097,362-400 does: blob
 * appended to the BIN chunk, zero-padded to 4 bytes (the bufferView's byteLength includes the pad), placeholder accessors, the extension's
 * attribute ids in add order, every other bufferView carried over, JSON chunk space-padded.  The primitives of ALL assets go through the device
 * together: one dmi_transcoder per entry of `devices` (NULL / 0: cfg->device), the least loaded one takes the next primitive; files are written
 * by library threads as their last primitive becomes final, into memory the result owns.  Inputs that lie in dmi_host_alloc memory go up without
 * a host pack (above).  `flags`: 0.
 * JSON byte equality with the reference is not part of the bit-exact contract; the embedded blobs are. */
typedef struct dmi_span { const uint8_t* data; size_t bytes; } dmi_span;
typedef struct dmi_gltf_asset {
  const uint8_t* glb; size_t glb_bytes;          /* a GLB container, or (glb == NULL) */
  const char* json; size_t json_bytes;           /* a glTF document and */
  const dmi_span* buffers; uint32_t n_buffers;   /* the bytes of its buffers, in `buffers` order */
} dmi_gltf_asset;
typedef struct dmi_transcode_stats {
  uint32_t files, primitives, devices;
  uint32_t buffers_in_place;   /* input buffers that lie in dmi_host_alloc memory (their accessors go up without a host pack) */
  uint32_t primitives_device_built, primitives_host_built, primitives_in_place, pad;   /* dmi_transcoder_counts, summed over the devices */
  uint64_t triangles_in, bytes_in, bytes_out;
  double parse_ms;      /* containers, JSON, primitive plans, accessor descriptors: wall clock from the start of the call to the last file parsed (a thread pool; pushes run beside it) */
  double pushed_ms;     /* since the start of the call: the last primitive handed to a transcoder */
  double finished_ms;   /* since the start of the call: the last stage of the last device coded */
  double build_ms, prepare_ms, encode_ms;   /* time inside the three stage calls, summed over stages and devices (they overlap) */
  double assemble_ms;   /* assembly threads, summed */
  double call_ms;
  double parse_cpu_ms;  /* round 6: the parse runs on a pool of threads — parse_ms is its wall-clock span (start of the call → last file parsed), this the threads' summed time */
  uint32_t parse_threads;
  uint32_t stages;      /* pipeline stages dispatched, summed over the devices */
} dmi_transcode_stats;
typedef struct dmi_transcoded dmi_transcoded;
int dmi_transcode_assets(const dmi_gltf_asset* assets, uint32_t n, const dmi_config* cfg, const int32_t* devices, uint32_t n_devices, uint32_t flags, dmi_transcoded** out);
/* file i of the result: its GLB bytes (the result's memory, valid until dmi_transcoded_free) and how many primitives were compressed;
 * dmi_transcoded_blobs: where their blobs lie in the file (offset, size without the pad), in primitive order */
int dmi_transcoded_file(const dmi_transcoded* r, uint32_t i, const uint8_t** glb, size_t* bytes, uint32_t* n_blobs);
int dmi_transcoded_blobs(const dmi_transcoded* r, uint32_t i, uint64_t* offsets, uint64_t* sizes);
/* the same for ALL files in one call: per file its address, size and blob count (arrays of stats.files entries); the blobs' (offset, size) pairs
 * back to back in file order (stats.primitives entries at most: blob_capacity = room in the two arrays) */
int dmi_transcoded_table(const dmi_transcoded* r, uint64_t* file_address, uint64_t* file_bytes, uint32_t* file_blobs, uint64_t* blob_offsets, uint64_t* blob_sizes, uint64_t blob_capacity);
/* the blocks of the result's output arena (every file lies inside one): returns their number, fills at most `capacity` (address, bytes) entries */
uint32_t dmi_transcoded_blocks(const dmi_transcoded* r, uint64_t* address, uint64_t* bytes, uint32_t capacity);
int dmi_transcoded_stats(const dmi_transcoded* r, dmi_transcode_stats* s);
void dmi_transcoded_free(dmi_transcoded* r);
/* The JSON layer of the above on its own (host only, for tests): parse `text`, write it back compactly — members in document order, number tokens
 * as written.  out: dmi_free. */
int dmi_json_roundtrip(const char* text, size_t n, dmi_buffer* out);

/* Stage times of the calling thread's last dmi_meshes_build (milliseconds; kernels_ms is hipEvent time summed over the groups). */
typedef struct dmi_build_timings {
  float pack_ms;       /* host threads: rows and indices into pinned staging */
  float kernels_ms;    /* device: the build kernels of all groups */
  float call_ms;       /* the whole call */
  uint32_t device_meshes, host_meshes;   /* primitives built by the kernels / by the host builder */
  uint64_t bytes_up, bytes_down;
  uint32_t in_place_meshes;              /* of device_meshes: accessors and indices read where they lie (dmi_host_register), nothing packed */
  uint32_t pad;
} dmi_build_timings;
int dmi_last_build_timings(dmi_build_timings* t);

/* --- Decoder side: the attribute section read back --------------------------------------------------------------
 * What a decoder does after its connectivity stage: `tables[i]` / `seeds` are the arrays dmi_encode_attributes takes (a decoder
 * rebuilds exactly these from the connectivity bytes), `section` is the output of dmi_encode_attributes / dmi_job_encode.
 * Entropy decoding (decode/entropy/{rans,symbol_coding}.rs) and the predictions that need earlier values of the same attribute run
 * on host cores — they are dependency chains like their encoders — normals (predicted from the decoded positions) and all
 * dequantization run on the device.  The reference's own decode/attribute/ is an unbuilt prototype of an older layout
 * (decode/mod.rs:5-6); everything above the entropy layer is the encoder spec inverted.
 * values: num_points rows of num_components f32 (the raw 4-byte values of a ToBits attribute), library-owned until dmi_decoded_free. */
typedef struct dmi_decoded_attribute {
  uint8_t att_type, component_type, num_components, domain;
  uint8_t scheme, transform, portabilization, bits;   /* wire ids of the prediction scheme / transform / portabilization, quantization bits */
  uint32_t unique_id;
  uint32_t num_points;
  const float* values;
} dmi_decoded_attribute;
typedef struct dmi_decoded { uint32_t num_attributes; const dmi_decoded_attribute* attributes; void* owner; } dmi_decoded;
int dmi_decode_attributes(const uint8_t* section, size_t len, const dmi_corner_table* tables, uint32_t n_tables, const uint32_t* seeds, uint32_t n_seeds,
                          uint32_t num_points, const dmi_config* cfg, dmi_decoded* out);
void dmi_decoded_free(dmi_decoded* d);

/* The slot order of the tile-sorted quantize gather (job creation of large meshes) on its own, for tests and tuning: inside every
 * tile of tile_entries consecutive sequence entries (rounded up to a power of two) the slots are ordered by point index — slot j reads point
 * slot_point[j] and writes sequence entry slot_entry[j]; block_entries = what one workgroup sorts in LDS (≤ 16384; larger tiles run their long
 * strides in global memory).  No reference counterpart: a layout choice of this library; the bitstream does not depend on it. */
int dmi_tile_sort_slots(const uint32_t* sequence_to_point, uint32_t n, uint32_t tile_entries, uint32_t block_entries, const dmi_config* cfg,
                        uint32_t* slot_point, uint32_t* slot_entry);

/* A whole `.drc` read back from its bytes alone (round 3): header (encode/header/mod.rs:26-54), Edgebreaker connectivity in the standard
 * traversal (the format encode/connectivity/edgebreaker.rs:458-656 writes, decoded the way every Draco-family decoder does: symbols in stored
 * order, one face per symbol glued to the open boundary, topology splits, interior start faces, attribute seams → per-attribute corner
 * tables), then dmi_decode_attributes on the rebuilt tables.  The reference's own decoder is not part of its crate (lib.rs:14) and its
 * connectivity decoder is unimplemented (decode/connectivity/spirale_reversi.rs:1088); this is the inverse of the ENCODER's format.
 * faces: 3·num_faces point indices in decode order (the reverse of the coding order); points: corners that agree in the universal vertex
 * and in every attribute's vertex; attributes[i].values: num_points rows.  Everything is library-owned until dmi_decoded_mesh_free. */
/* Host wall clock of the calling thread's last dmi_decode_attributes / dmi_decode_mesh, by stage. */
typedef struct dmi_decode_timings {
  float connectivity_ms;   /* dmi_decode_mesh only: symbols → universal corner table + seam flags */
  float tables_ms;         /* dmi_decode_mesh only: per-attribute corner tables from the seams + point ids */
  float sequence_ms;       /* attribute traversals (one host thread per distinct corner table), run BESIDE the entropy decoders */
  float entropy_ms;        /* traversals + rANS symbols + rABS bits, one host thread per attribute (decode/entropy/rans.rs:36-69): the wall
                              clock of both together (sequence_ms is contained in it) */
  float inverse_ms;        /* sequential predictions inverted on a host core (positions, texture coordinates, generic) */
  float device_ms;         /* uploads, k_decode_normals, k_dequantize, read-back */
  float attributes_ms;     /* the whole dmi_decode_attributes */
  float call_ms;           /* dmi_decode_mesh: the whole call */
} dmi_decode_timings;
int dmi_last_decode_timings(dmi_decode_timings* t);

typedef struct dmi_decoded_mesh {
  uint32_t num_faces, num_points;
  const uint32_t* faces;
  uint32_t num_attributes;
  const dmi_decoded_attribute* attributes;
  void* owner;
} dmi_decoded_mesh;
int dmi_decode_mesh(const uint8_t* drc, size_t len, const dmi_config* cfg, dmi_decoded_mesh* out);
/* The connectivity half alone (host only, no GPU): header + connectivity bytes → the corner tables and seeds dmi_decode_attributes takes
 * (tables[j].sequence stays null: the decoder derives it).  *consumed (nullable) = where the attribute section starts. */
int dmi_decode_connectivity(const uint8_t* header_and_connectivity, size_t len, dmi_conn* conn, size_t* consumed);
void dmi_decoded_conn_free(dmi_conn* conn);
void dmi_decoded_mesh_free(dmi_decoded_mesh* m);

/* --- The hybrid form's host-core stream coders on their own (host only, no device) ----------------
 * A single large mesh codes its streams on host cores from the device-built symbols and tables (dmi_job_encode, see
 * DESIGN.md §5); these two entry points expose exactly those coders so that tests can pin them against the reference's
 * RansCoder / RabsCoder (encode/entropy/rans.rs:33-68, :91-128) and the bench can time one host core beside the device
 * walker.  freq[] = normalised frequencies summing to 2^precision; symbols are coded LAST TO FIRST
 * (encode/entropy/symbol_coding.rs:161-163), bits first to last; the tagged final state is appended (rans.rs:48-68). */
int dmi_host_rans_stream(const uint32_t* freq, uint32_t num_symbols, uint32_t precision, const uint32_t* symbols, uint64_t n, dmi_buffer* out);
int dmi_host_rabs_stream(uint8_t zero_prob, const uint8_t* bits, uint64_t n, dmi_buffer* out);
/* The same stream for n copies of ONE bit without stepping through them: with the bit fixed the coder's states repeat (zero_prob 255: after
 * 1410 steps, period 1409 steps / 1 byte), so the bytes are a prefix, the period's bytes repeated, a tail.  The connectivity stage codes
 * the seam flags of every attribute without seams this way (encode/connectivity/edgebreaker.rs:611-653: 1.5 zero flags per face). */
int dmi_host_rabs_constant_stream(uint8_t zero_prob, uint32_t bit, uint64_t n, dmi_buffer* out);

void dmi_free(dmi_buffer* buf);
/* dmi_free of bufs[0..n): the outputs of a batch call released in one call */
void dmi_free_many(dmi_buffer* bufs, uint32_t n);
const char* dmi_strerror(int status);
/* Last error detail for the calling thread (HIP error string, offending attribute, ...). */
const char* dmi_last_error(void);
/* Host threads the library may use in one call: the machine's hardware threads, no more than the cgroup's CPU quota, DMI_HOST_THREADS, and — for
 * calls made by the CALLING thread — the cap set here (0 = none).  A pipeline that runs several library calls side by side (a transcode: build ∥
 * prepare ∥ encode) gives each stage's thread its share, so that together they stay inside the quota. */
void dmi_thread_host_threads(uint32_t n);
int dmi_usable_host_threads(void);
/* Number of HIP devices visible (0 when there is no GPU); never initialises a context. */
int dmi_device_count(void);
/* Optional: pay a process's one-time costs for `device` now instead of inside its first encode — HIP context, the library's code objects
 * (one launch per kernel file), the calling thread's library stream, `staging_bytes` of pinned huge-page staging and `device_bytes` of
 * pooled device memory (both 0 = none; a 10M-triangle whole-mesh call uses ≈ 250 MB and ≈ 4 GB).  The first whole-mesh call of a process
 * is otherwise 0.15–0.2 s slower than the following ones. */
int dmi_init(int device, size_t staging_bytes, size_t device_bytes);
/* The library keeps released device chunks, idle pinned staging buffers and the large host arrays of the connectivity stage for its next
 * call (the host arrays up to DMI_HOST_CACHE_MB, default 4096; 0 = keep none).  This hands all of it back; live jobs are untouched. */
void dmi_release_cached_memory(void);

#ifdef __cplusplus
}
#endif
#endif /* DRACO_MI_H */
