// dmi_prepare.hpp — what the whole-mesh translation units share (round 5 split of dmi_prepare.cpp: single-mesh entry points there, the batch
// forms in dmi_prepare_batch.cpp, library streams / NUMA placement in dmi_streams.cpp).
#pragma once
#include "dmi_job.hpp"

namespace dmi {

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

// The universal corner table of a mesh as the device connectivity stage built it, read back for the host's serial walks (pinned staging).
struct PrebuiltTable {
  const uint32_t *c2v = nullptr, *opp = nullptr, *lmc = nullptr;
  const uint8_t* on_boundary = nullptr;
  uint32_t V = 0;
  bool no_boundary = false;
  bool quad = false;   // `opp` holds 4·face + k ids (CornerTables::quad)
  // attribute corner tables the device built (k_att_*), by index among the mesh's non-position attributes; ready = false: the host builds it
  struct Att { bool ready = false, interior = false; uint32_t nv = 0; const uint8_t* seam = nullptr; const uint32_t *c2v = nullptr, *opp = nullptr, *lmc = nullptr;
               const uint32_t *d_c2v = nullptr, *d_opp = nullptr; };
  std::vector<Att> att;
};

constexpr uint32_t kDeviceTablesMinFaces = 1u << 16;   // a single mesh from this size up gets its universal corner table from the device (dmi_conn.hip)

// The universal corner table of ONE mesh built on the device (dmi_conn.hip) and read back for the host's serial walks: faces up,
// opposite corners + per-vertex boundary flags down (pinned staging), the device copies kept for job creation (coding-order relabelling

// ------------------------------------------------------------------------------------------------
// Whole-mesh entry points: connectivity stage (host walks, device tables) + device attributes.
// ------------------------------------------------------------------------------------------------
struct ConnOwner {
  CornerTables ct;
  EdgebreakerResult eb;
  std::vector<std::vector<uint32_t>> seqs;
  ConnOwner() = default;
  ConnOwner(const ConnOwner&) = delete;
  ConnOwner& operator=(const ConnOwner&) = delete;
  ~ConnOwner() {   // the large arrays go back to the host pool (dmi_host.hpp)
    pool_give(ct.c2p_own); pool_give(ct.c2v_own); pool_give(ct.opp_own); pool_give(ct.lmc_own);
    for (auto& a : ct.att) { pool_give(a.c2v); pool_give(a.opp); pool_give(a.lmc); pool_give(a.seam_edge); }
    pool_give(eb.seeds); pool_give(eb.processed);
    for (auto& q : seqs) pool_give(q);
  }
  std::vector<dmi_corner_table> views;
};

// thread time of the connectivity stage by step, summed over the meshes of a batch (trace): attribute tables, Edgebreaker (traversal +

// thread time of the connectivity stage by step, summed over the meshes of a batch (trace): attribute tables, Edgebreaker (traversal +
// connectivity bytes incl. the seam streams), universal sequencer, seam-table sequencers
extern std::atomic<uint64_t> g_conn_us[4];
// set by a batch worker whose batch keeps every host thread busy with a mesh of its own: a large mesh then walks its steps one after the other on
// its worker (the overlapped form starts three more threads per mesh — with 16 workers on 16 CPUs they only wait for each other)
extern thread_local bool g_batch_worker_busy;
// pre (nullable): the universal table already built by the device stage; view_faces: c2p may view the caller's face array (it outlives `o`)
int build_connectivity(const dmi_mesh* mesh, ConnOwner& o, std::vector<uint8_t>& bytes, const PrebuiltTable* pre = nullptr, bool view_faces = false);
// the faces / position map of a mesh already in device memory (dmi_encode_mesh_device, a large member of a built group)
struct DeviceMeshSrc { const uint32_t* faces; const uint32_t* pos_map; };
constexpr int kNeedHostValues = -77;   // mesh_prepare_impl: such a mesh has to take the host builders (its values must come down first)
int mesh_prepare_impl(const dmi_mesh* mesh, const dmi_config* cfg, dmi_buffer* header_and_connectivity, dmi_job** job, const DeviceMeshSrc* src);

// the attribute tables the device built for member `member` of a group (k_att_*), by index among the mesh's non-position attributes
void att_stage_fill(const AttStage& st, uint32_t member, uint32_t n_nonpos, std::vector<PrebuiltTable::Att>& out);

}  // namespace dmi
