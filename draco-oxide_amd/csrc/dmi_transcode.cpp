// dmi_transcode.cpp — the transcoder's per-primitive loop behind ONE object: primitives are pushed as the caller's importer produces them, and
// stages of them run build → prepare → encode on library threads — two per step since round 5 — (dmi_meshes_build of stage k+2 beside dmi_built_meshes_prepare of stage
// k+1 beside dmi_jobs_encode of stage k); a callback names the primitives whose blobs are final, so that the caller reassembles files while the
// device works on the next ones.
// Reference seam: io/gltf/transcoder.rs:134-151 (files one by one), io/gltf/encode.rs:932-955,1827-1842 (their primitives one by one:
// MeshBuilder::build → encode::encode → bufferView).  Same bytes per primitive as dmi_mesh_build + dmi_encode_mesh (tests/test_gltf.py).
// Why inside the library: the three calls of a stage were glued together by the caller's interpreter (job wrappers, byte copies, queue hand-overs
// under one interpreter lock: ≈ 120 of a 165 ms transcode of 1024 files); here a stage changes hands without it.
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <condition_variable>
#include <deque>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/draco_mi.h"
#include "dmi_host.hpp"

using namespace dmi;

namespace {

struct Stage {
  uint32_t first = 0, count = 0;            // primitives [first, first + count) in push order
  std::vector<dmi_raw_mesh> raw;            // their descriptors (a copy: the transcoder's own list may grow — and move — while the stage is built)
  std::vector<dmi_built_mesh> built;        // count entries (freed by the prepare step)
  std::vector<uint32_t> kept;               // indices (relative to first) of the primitives with a face left
  std::vector<dmi_job*> jobs;               // one per kept primitive (destroyed by the encode step)
};

// a bounded hand-over between two steps (one stage in flight between neighbours, like the Python driver's queues)
struct Slot {
  std::mutex m;
  std::condition_variable cv;
  std::deque<std::unique_ptr<Stage>> q;
  bool closed = false;
  size_t cap = 1;
  bool put(std::unique_ptr<Stage> s) {
    std::unique_lock<std::mutex> lock(m);
    cv.wait(lock, [&] { return q.size() < cap || closed; });
    if (closed) return false;
    q.push_back(std::move(s));
    cv.notify_all();
    return true;
  }
  std::unique_ptr<Stage> take() {   // null: closed and drained
    std::unique_lock<std::mutex> lock(m);
    cv.wait(lock, [&] { return !q.empty() || closed; });
    if (q.empty()) return nullptr;
    std::unique_ptr<Stage> s = std::move(q.front());
    q.pop_front();
    cv.notify_all();
    return s;
  }
  void close() { std::lock_guard<std::mutex> lock(m); closed = true; cv.notify_all(); }
};

}  // namespace

struct dmi_transcoder {
  dmi_config cfg{};
  dmi_debug debug{};                   // the switches of the call that created it (cfg.debug points here: every stage call of its threads opens its scope with them)
  uint64_t stage_triangles = 0;
  dmi_transcode_done_fn done = nullptr;
  void* user = nullptr;
  // what was pushed (descriptors are copied; the arrays they point to stay the caller's until the primitive is done)
  std::mutex push_mutex;
  std::deque<std::vector<dmi_raw_accessor>> accessors;   // (stable addresses)
  std::vector<dmi_raw_mesh> prims;
  uint32_t dispatched = 0;             // primitives already handed to the build step
  uint32_t stages = 0;                 // stages dispatched so far
  uint32_t stage_primitives = 0;       // 0 = no cap on the primitives of a stage
  uint32_t stage_ramp = 0;             // n > 0: stage k takes min(1, 2^k / n) of stage_triangles
  uint64_t pending_triangles = 0;
  bool first_stage = true;
  // results, by primitive
  std::vector<dmi_buffer> heads, sections;
  std::vector<uint32_t> num_faces, num_points;
  std::mutex result_mutex;
  // steps
  Slot to_build, to_prepare, to_encode;
  dmi::Thread t_build, t_build2, t_prepare, t_prepare2, t_prepare3, t_prepare4, t_encode, t_encode2;
  uint32_t prepare_threads = 2;
  std::atomic<int> builders_left{0}, preparers_left{0};
  std::mutex err_mutex;
  int rc = DMI_OK;
  std::string err;
  double ms_build = 0, ms_prepare = 0, ms_encode = 0;
  uint64_t n_device = 0, n_host = 0, n_in_place = 0;   // primitives built by the kernels / by the host builder / of the former: copied up where they lay
  bool trace = false;   // (DMI_DBG_TRACE_STAGES: only the stage lines below — the full trace prints a line per mesh)
  const double t_create = now_ms();
  void note(const char* step, const Stage& s, double t0) const { if (trace) std::fprintf(stderr, "[dmi] transcoder %-8s stage@%-5u (%4u primitives) %7.1f -> %7.1f ms\n", step, s.first, s.count, t0 - t_create, now_ms() - t_create); }
  bool started = false, finished = false;

  void fail_with(int code) {
    std::lock_guard<std::mutex> lock(err_mutex);
    if (rc == DMI_OK) { rc = code; err = dmi_last_error(); }
  }
  bool failed() { std::lock_guard<std::mutex> lock(err_mutex); return rc != DMI_OK; }

  static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

  void build_loop() {
    while (std::unique_ptr<Stage> s = to_build.take()) {
      if (failed()) continue;
      const double t0 = now_ms();
      s->built.assign(s->count, dmi_built_mesh{});
      const int r = dmi_meshes_build(s->raw.data(), s->count, &cfg, 0u, s->built.data());
      dmi_build_timings bt{};
      (void)dmi_last_build_timings(&bt);
      { std::lock_guard<std::mutex> lock(err_mutex); ms_build += now_ms() - t0; if (!r) { n_device += bt.device_meshes; n_host += bt.host_meshes; n_in_place += bt.in_place_meshes; } }
      note("build", *s, t0);
      if (r) { fail_with(r); continue; }
      if (!to_prepare.put(std::move(s))) break;
    }
    if (builders_left.fetch_sub(1) == 1) to_prepare.close();   // (the last build thread out)
  }
  void prepare_loop() {
    while (std::unique_ptr<Stage> s = to_prepare.take()) {
      const double t0 = now_ms();
      int r = DMI_OK;
      if (!failed()) {
        std::vector<uint32_t> nf(s->count), np(s->count);
        r = dmi_built_meshes_info(s->built.data(), s->count, nf.data(), np.data());
        std::vector<dmi_built_mesh> sel;
        if (!r) {
          for (uint32_t k = 0; k < s->count; ++k) if (nf[k]) { s->kept.push_back(k); sel.push_back(s->built[k]); }
          std::lock_guard<std::mutex> lock(result_mutex);
          for (uint32_t k = 0; k < s->count; ++k) { num_faces[s->first + k] = nf[k]; num_points[s->first + k] = nf[k] ? np[k] : 0; }
        }
        if (!r && !sel.empty()) {
          std::vector<dmi_buffer> h(sel.size());
          s->jobs.assign(sel.size(), nullptr);
          r = dmi_built_meshes_prepare(sel.data(), (uint32_t)sel.size(), &cfg, h.data(), s->jobs.data());
          if (!r) { std::lock_guard<std::mutex> lock(result_mutex); for (size_t q = 0; q < sel.size(); ++q) heads[s->first + s->kept[q]] = h[q]; }
          else s->jobs.clear();
        }
      }
      dmi_built_meshes_free(s->built.data(), s->count);   // (the jobs copied what they need)
      s->built.clear();
      { std::lock_guard<std::mutex> lock(err_mutex); ms_prepare += now_ms() - t0; }
      note("prepare", *s, t0);
      if (r) { fail_with(r); continue; }
      if (failed()) { for (dmi_job* j : s->jobs) dmi_job_destroy(j); continue; }
      if (!to_encode.put(std::move(s))) break;
    }
    if (preparers_left.fetch_sub(1) == 1) to_encode.close();   // (the last prepare thread out)
  }
  void encode_loop() {
    while (std::unique_ptr<Stage> s = to_encode.take()) {
      const double t0 = now_ms();
      int r = DMI_OK;
      if (!failed() && !s->jobs.empty()) {
        std::vector<dmi_buffer> outs(s->jobs.size());
        r = dmi_jobs_encode(s->jobs.data(), (uint32_t)s->jobs.size(), outs.data());
        if (!r) { std::lock_guard<std::mutex> lock(result_mutex); for (size_t q = 0; q < outs.size(); ++q) sections[s->first + s->kept[q]] = outs[q]; }
        else dmi_free_many(outs.data(), (uint32_t)outs.size());   // (what a failed batch had already produced)
      }
      for (dmi_job* j : s->jobs) dmi_job_destroy(j);
      s->jobs.clear();
      { std::lock_guard<std::mutex> lock(err_mutex); ms_encode += now_ms() - t0; }
      note("encode", *s, t0);
      if (r) { fail_with(r); continue; }
      if (!failed() && done) done(user, s->first, s->count);
    }
  }
  void start() {
    if (started) return;
    started = true;
    // Two threads per step.  Builds: the packing of stage k+1 beside the upload / kernels / read-back of stage k.  Prepares: stage k+1's walks start
    // while stage k's coordinator waits for the device (sequences up, relabelling, fan rows: 3–10 ms per stage with every host core idle); the
    // walks of both share one process-wide budget (WalkSlots).  Encodes: a stage's chain launch is bounded by its longest stream (≈ 5 ms on one
    // wavefront per SIMD, the rest of the chip idle) — the launches of consecutive stages overlap instead of queueing at the end of the call.
    // (1024 files, one thread for one of the steps: 100–125 ms against 86–97.)
    builders_left = 2;
    t_build = dmi::Thread([this] { build_loop(); });
    t_build2 = dmi::Thread([this] { build_loop(); });
    preparers_left = (int)prepare_threads;
    t_prepare = dmi::Thread([this] { prepare_loop(); });
    if (prepare_threads > 1) t_prepare2 = dmi::Thread([this] { prepare_loop(); });
    if (prepare_threads > 2) t_prepare3 = dmi::Thread([this] { prepare_loop(); });
    if (prepare_threads > 3) t_prepare4 = dmi::Thread([this] { prepare_loop(); });
    t_encode = dmi::Thread([this] { encode_loop(); });
    t_encode2 = dmi::Thread([this] { encode_loop(); });
  }
  // hands the primitives pushed so far to the build step once they make a stage (or all of them: flush)
  void dispatch(bool flush) {
    // (the first stage runs alone: the sooner it is through, the sooner the steps overlap.  Small LAST stages — half of what is expected to be left,
    // down to a third of a stage — were tried against the tail of the call: no gain beyond the noise, 105–110 against 95–105 ms per 1024 files)
    const uint64_t want = stage_ramp ? std::max<uint64_t>(1, std::min<uint64_t>(stage_triangles, (stage_triangles / stage_ramp) << std::min<uint32_t>(stages, 16u)))
                                     : first_stage ? std::max<uint64_t>(1, stage_triangles / 3) : stage_triangles;
    // (… or stage_primitives of them: with the files taken largest first the last stage of a long list would otherwise hold most of its primitives —
    //  695 of 1024 — and its build, prepare and encode, which nothing overlaps, pay per primitive)
    const bool many = stage_primitives && prims.size() - dispatched >= stage_primitives;
    if (dispatched == prims.size() || (!flush && pending_triangles < want && !many)) return;
    std::unique_ptr<Stage> s(new Stage());
    s->first = dispatched; s->count = (uint32_t)prims.size() - dispatched;
    s->raw.assign(prims.begin() + dispatched, prims.end());
    dispatched = (uint32_t)prims.size();
    ++stages;
    pending_triangles = 0;
    first_stage = false;
    start();
    (void)to_build.put(std::move(s));
  }
};

extern "C" {

dmi_transcoder* dmi_transcoder_create(const dmi_config* cfg, uint64_t expected_triangles, uint64_t stage_triangles, dmi_transcode_done_fn done, void* user) {
  std::unique_ptr<dmi_transcoder> t(new dmi_transcoder());
  DebugScope scope(cfg ? cfg->debug : nullptr);
  if (cfg) t->cfg = *cfg;
  t->debug = dbg();
  t->cfg.debug = &t->debug;
  t->trace = dbg_on(DMI_DBG_TRACE | DMI_DBG_TRACE_STAGES);
  t->stage_ramp = dbg().stage_ramp;
  t->prepare_threads = dbg().prepare_threads ? std::min<uint32_t>(4u, dbg().prepare_threads) : 2u;
  t->stage_primitives = dbg().stage_primitives ? dbg().stage_primitives : 256u;   // (1024 files, medians of 7: 80.5 → 80.1, 87.6 → 84.9, 88.4 → 80.8 ms with the cap)
  // about four stages (enough to overlap the steps), between 3M and 12M triangles: a stage pays fixed costs (the chain launch of its encode is bounded
  // by its longest stream, ≈ 5 ms) and one above ≈ 16M stops overlapping (measured with the Python driver: DESIGN §6b)
  t->stage_triangles = stage_triangles ? stage_triangles : std::min<uint64_t>((uint64_t)12 << 20, std::max<uint64_t>((uint64_t)3 << 20, expected_triangles / 4));
  t->done = done; t->user = user;
  t->to_build.cap = 2;   // (the caller may run a stage ahead of the build)
  return t.release();
}

int dmi_transcoder_push(dmi_transcoder* t, const dmi_raw_mesh* prims, uint32_t n) {
  if (!t || (!prims && n)) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  if (t->finished) return host_fail(DMI_ERR_INVALID_ARGUMENT, "transcoder already finished");
  if (t->failed()) { std::lock_guard<std::mutex> lock(t->err_mutex); return host_fail(t->rc, t->err); }
  {
    std::lock_guard<std::mutex> lock(t->push_mutex);
    for (uint32_t k = 0; k < n; ++k) {
      t->accessors.emplace_back(prims[k].atts, prims[k].atts + prims[k].n_atts);
      dmi_raw_mesh m = prims[k];
      m.atts = t->accessors.back().data();
      t->prims.push_back(m);
      t->pending_triangles += prims[k].num_faces;
    }
    {
      std::lock_guard<std::mutex> rlock(t->result_mutex);
      t->heads.resize(t->prims.size(), dmi_buffer{}); t->sections.resize(t->prims.size(), dmi_buffer{});
      t->num_faces.resize(t->prims.size(), 0); t->num_points.resize(t->prims.size(), 0);
    }
    t->dispatch(false);   // (under the push mutex: the primitive list, the dispatch cursor and the thread start belong to one pusher at a time)
  }
  return DMI_OK;
}

int dmi_transcoder_reserve(dmi_transcoder* t, uint32_t n_primitives) {
  if (!t) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  std::lock_guard<std::mutex> lock(t->push_mutex);
  t->prims.reserve(n_primitives);   // (a hint: a stage carries a copy of its descriptors, the list may grow at any time)
  std::lock_guard<std::mutex> rlock(t->result_mutex);
  t->heads.reserve(n_primitives); t->sections.reserve(n_primitives); t->num_faces.reserve(n_primitives); t->num_points.reserve(n_primitives);
  return DMI_OK;
}

int dmi_transcoder_finish(dmi_transcoder* t) {
  if (!t) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  if (!t->finished) {
    { std::lock_guard<std::mutex> lock(t->push_mutex); t->dispatch(true); }
    t->to_build.close();
    if (t->t_build.joinable()) t->t_build.join();
    if (t->t_build2.joinable()) t->t_build2.join();
    if (t->t_prepare.joinable()) t->t_prepare.join();
    if (t->t_prepare2.joinable()) t->t_prepare2.join();
    if (t->t_prepare3.joinable()) t->t_prepare3.join();
    if (t->t_prepare4.joinable()) t->t_prepare4.join();
    if (t->t_encode.joinable()) t->t_encode.join();
    if (t->t_encode2.joinable()) t->t_encode2.join();
    t->finished = true;
  }
  std::lock_guard<std::mutex> lock(t->err_mutex);
  return t->rc ? host_fail(t->rc, t->err) : DMI_OK;
}

int dmi_transcoder_result(dmi_transcoder* t, uint32_t i, dmi_buffer* header_and_connectivity, dmi_buffer* section, uint32_t* num_faces, uint32_t* num_points) {
  if (!t) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  std::lock_guard<std::mutex> lock(t->result_mutex);
  if (i >= t->heads.size()) return host_fail(DMI_ERR_INVALID_ARGUMENT, "primitive index out of range");
  if (header_and_connectivity) *header_and_connectivity = t->heads[i];
  if (section) *section = t->sections[i];
  if (num_faces) *num_faces = t->num_faces[i];
  if (num_points) *num_points = t->num_points[i];
  return DMI_OK;
}

int dmi_transcoder_timings(dmi_transcoder* t, double* build_ms, double* prepare_ms, double* encode_ms) {
  if (!t) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  if (build_ms) *build_ms = t->ms_build;
  if (prepare_ms) *prepare_ms = t->ms_prepare;
  if (encode_ms) *encode_ms = t->ms_encode;
  return DMI_OK;
}

uint32_t dmi_transcoder_stages(dmi_transcoder* t) {
  if (!t) return 0;
  std::lock_guard<std::mutex> lock(t->push_mutex);
  return t->stages;
}

int dmi_transcoder_counts(dmi_transcoder* t, uint64_t* device_built, uint64_t* host_built, uint64_t* in_place) {
  if (!t) return host_fail(DMI_ERR_INVALID_ARGUMENT, "null");
  std::lock_guard<std::mutex> lock(t->err_mutex);
  if (device_built) *device_built = t->n_device;
  if (host_built) *host_built = t->n_host;
  if (in_place) *in_place = t->n_in_place;
  return DMI_OK;
}

void dmi_transcoder_destroy(dmi_transcoder* t) {
  if (!t) return;
  if (!t->finished) (void)dmi_transcoder_finish(t);
  for (auto& b : t->heads) if (b.data) dmi_free(&b);       // (the buffers stay the transcoder's: views of them are valid until here)
  for (auto& b : t->sections) if (b.data) dmi_free(&b);
  delete t;
}

}  // extern "C"
