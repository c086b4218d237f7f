// dmi_batch.cpp — batches of resident jobs: dmi_jobs_encode / dmi_jobs_encode_devices.  Every job's data-parallel phases, table stage and
// record prep are planned together (one upload, one launch per (level, kernel)), all rANS/rABS streams of all jobs run in ONE persistent
// chain launch, and the coded bytes come back in one packed read-back.
#include "dmi_job.hpp"

using namespace dmi;

namespace {
// the chain launches of batch encodes in flight per device (see jobs_encode_impl)
struct ChainSlots { std::mutex m; std::condition_variable cv; int in_use[64] = {0}; };
ChainSlots& chain_slots() { static ChainSlots c; return c; }
struct ChainSlot {
  int d;
  explicit ChainSlot(int device) : d(device & 63) { ChainSlots& c = chain_slots(); std::unique_lock<std::mutex> lock(c.m); c.cv.wait(lock, [&] { return c.in_use[d] < 2; }); ++c.in_use[d]; }
  ~ChainSlot() { ChainSlots& c = chain_slots(); { std::lock_guard<std::mutex> lock(c.m); --c.in_use[d]; } c.cv.notify_one(); }
  ChainSlot(const ChainSlot&) = delete;
  ChainSlot& operator=(const ChainSlot&) = delete;
};
}  // namespace

// dmi_jobs_encode calls do not share one).
struct BatchArena {
  int device = -1;
  void* bytes_dev = nullptr; size_t bytes_dev_cap = 0;
  void* table_dev = nullptr; void* table_host = nullptr; size_t table_cap = 0;
  void* bytes_host = nullptr; size_t bytes_host_cap = 0;
  void* plan_host = nullptr; void* plan_dev = nullptr; size_t plan_cap = 0;       // launch plan of a batch (argument blocks, block maps)
  void* slabs_host = nullptr; void* slabs_dev = nullptr; size_t slabs_cap = 0;   // every job's slab, packed
  void* descs_dev = nullptr; size_t descs_cap = 0;   // device form: chain descriptors | header pseudo-descriptors | stream order | pull counter
  int reserve_descs(size_t bytes) {
    if (bytes <= descs_cap) return DMI_OK;
    if (descs_dev) (void)hipFree(descs_dev);
    descs_dev = nullptr; descs_cap = 0;
    HIP_TRY(hipMalloc(&descs_dev, bytes + bytes / 4 + 4096));
    descs_cap = bytes + bytes / 4 + 4096;
    return DMI_OK;
  }
  bool in_use = false;
  int reserve(size_t dev_bytes, size_t table_bytes) {
    if (dev_bytes > bytes_dev_cap) {
      if (bytes_dev) (void)hipFree(bytes_dev);
      bytes_dev = nullptr; bytes_dev_cap = 0;
      HIP_TRY(hipMalloc(&bytes_dev, dev_bytes + dev_bytes / 4 + 4096));
      bytes_dev_cap = dev_bytes + dev_bytes / 4 + 4096;
    }
    if (table_bytes > table_cap) {
      if (table_dev) (void)hipFree(table_dev);
      if (table_host) (void)hipHostFree(table_host);
      table_dev = table_host = nullptr; table_cap = 0;
      HIP_TRY(hipMalloc(&table_dev, table_bytes * 2));
      HIP_TRY(hipHostMalloc(&table_host, table_bytes * 2, hipHostMallocDefault));
      table_cap = table_bytes * 2;
    }
    return DMI_OK;
  }
  static int reserve_pair(void*& host, void*& dev, size_t& cap, size_t bytes) {
    if (bytes <= cap) return DMI_OK;
    if (host) (void)hipHostFree(host);
    if (dev) (void)hipFree(dev);
    host = dev = nullptr; cap = 0;
    const size_t want = bytes + bytes / 4 + 4096;
    HIP_TRY(hipHostMalloc(&host, want, hipHostMallocDefault));
    HIP_TRY(hipMalloc(&dev, want));
    cap = want;
    return DMI_OK;
  }
  int reserve_host(size_t bytes) {
    if (bytes > bytes_host_cap) {
      if (bytes_host) (void)hipHostFree(bytes_host);
      bytes_host = nullptr; bytes_host_cap = 0;
      HIP_TRY(hipHostMalloc(&bytes_host, bytes + bytes / 4 + 4096, hipHostMallocDefault));
      bytes_host_cap = bytes + bytes / 4 + 4096;
    }
    return DMI_OK;
  }
};
static std::mutex g_arena_mutex;
static std::vector<BatchArena*> g_arenas;   // (never freed: process-lifetime staging)
static BatchArena* acquire_batch_arena(int device) {
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  for (BatchArena* a : g_arenas) if (!a->in_use && a->device == device) { a->in_use = true; return a; }
  BatchArena* a = new BatchArena();
  a->device = device;
  a->in_use = true;
  g_arenas.push_back(a);
  return a;
}
static void release_batch_arena(BatchArena* a) {
  std::lock_guard<std::mutex> lock(g_arena_mutex);
  a->in_use = false;
}

// A batch's launch plan: the KernelSteps of many jobs grouped by (level, kernel); argument blocks, block maps and any extra
// tables go to the device in ONE copy, then every group is one multi-item launch.
struct BatchPlan {
  struct Group { int level = 0, id = 0; uint32_t lds = 0, total_blocks = 0; std::vector<const KernelStep*> items; size_t off_args = 0, off_info = 0, off_blocks = 0; };
  std::vector<Group> groups;
  size_t bytes = 0;
  static size_t align(size_t v) { return (v + 255) & ~(size_t)255; }
  void add(const std::vector<std::vector<KernelStep>>& steps, int n_levels) {
    // one pass: bucket (level, kernel) → group, in job order; then the groups in (level, kernel) order
    std::vector<Group> bucket((size_t)n_levels * K_COUNT);
    for (const auto& job_steps : steps)
      for (const KernelStep& st : job_steps) {
        if (st.level < 0 || st.level >= n_levels || st.id < 0 || st.id >= K_COUNT) continue;
        Group& g = bucket[(size_t)st.level * K_COUNT + st.id];
        g.items.push_back(&st); g.total_blocks += st.blocks; g.lds = std::max(g.lds, st.lds);
      }
    for (int level = 0; level < n_levels; ++level)
      for (int id = 0; id < K_COUNT; ++id) {
        Group& g = bucket[(size_t)level * K_COUNT + id];
        if (g.items.empty()) continue;
        g.level = level; g.id = id;
        groups.push_back(std::move(g));
      }
    for (Group& g : groups) {
      g.off_args = bytes; bytes = align(bytes + (size_t)g.items.size() * g.items[0]->args_size);
      g.off_info = bytes; bytes = align(bytes + (size_t)g.total_blocks * sizeof(uint2));
      g.off_blocks = bytes; bytes = align(bytes + g.items.size() * sizeof(uint32_t));
    }
  }
  size_t reserve(size_t n) { const size_t off = bytes; bytes = align(bytes + n); return off; }
  void fill_group(uint8_t* ph, const Group& g) const {
    const size_t asz = g.items[0]->args_size;
    uint2* info = reinterpret_cast<uint2*>(ph + g.off_info);
    uint32_t* blocks = reinterpret_cast<uint32_t*>(ph + g.off_blocks);
    uint32_t at = 0;
    for (size_t i = 0; i < g.items.size(); ++i) {
      std::memcpy(ph + g.off_args + i * asz, g.items[i]->args, asz);
      blocks[i] = g.items[i]->blocks;
      for (uint32_t b = 0; b < g.items[i]->blocks; ++b) info[at++] = make_uint2((uint32_t)i, b);
    }
  }
  void fill(uint8_t* ph) const {
    if (groups.size() < 4) { for (const Group& g : groups) fill_group(ph, g); return; }
    std::atomic<size_t> next{0};   // groups differ a lot in size: a few host threads pull them
    auto work = [&] { for (size_t k; (k = next.fetch_add(1)) < groups.size();) fill_group(ph, groups[k]); };
    std::vector<dmi::Thread> th;
    for (size_t t = 1; t < std::min<size_t>(groups.size(), 8); ++t) th.emplace_back(with_debug(work));
    work();
    for (auto& x : th) x.join();
  }
  void launch(const uint8_t* pd, hipStream_t s) const {
    for (const Group& g : groups)
      launch_steps_multi(g.id, pd + g.off_args, reinterpret_cast<const uint2*>(pd + g.off_info), reinterpret_cast<const uint32_t*>(pd + g.off_blocks), g.total_blocks, g.lds, s);
  }
};

// Phase A of a whole batch in ONE launch per (level, kernel): every job's launches are collected as KernelSteps (the same
// code path as a single encode, with a sink set), grouped, uploaded in one copy and served by multi-item kernels; the slabs
// come back packed in one copy.  Small meshes are otherwise bound by the ≈2.4 µs the GPU spends per tiny kernel (9 per job).
static int run_phase_a_batched(dmi_job** jobs, const std::vector<uint32_t>& which, BatchArena* arena, hipStream_t s) {
  const uint32_t n = (uint32_t)which.size();
  if (!n) return DMI_OK;
  std::vector<std::vector<KernelStep>> steps(n);
  for (uint32_t k = 0; k < n; ++k) {
    dmi_job* job = jobs[which[k]];
    set_step_sink(&steps[k]);
    const int rc = encode_phase_a(job, true);
    set_step_sink(nullptr);
    if (rc) return rc;
  }
  BatchPlan plan;
  plan.add(steps, kStepLevels);
  std::vector<CopyItem> copies(n);
  size_t slab_bytes = 0;
  for (uint32_t k = 0; k < n; ++k) {
    dmi_job* job = jobs[which[k]];
    copies[k] = CopyItem{job->slab.p, (uint64_t)slab_bytes, (uint64_t)(job->slab.bytes & ~(size_t)15)};
    slab_bytes = BatchPlan::align(slab_bytes + job->slab.bytes);
  }
  const size_t off_copies = plan.reserve(copies.size() * sizeof(CopyItem));
  int rc;
  if ((rc = BatchArena::reserve_pair(arena->plan_host, arena->plan_dev, arena->plan_cap, plan.bytes))) return rc;
  if ((rc = BatchArena::reserve_pair(arena->slabs_host, arena->slabs_dev, arena->slabs_cap, slab_bytes))) return rc;
  uint8_t* ph = static_cast<uint8_t*>(arena->plan_host);
  plan.fill(ph);
  std::memcpy(ph + off_copies, copies.data(), copies.size() * sizeof(CopyItem));
  HIP_TRY(hipMemcpyAsync(arena->plan_dev, arena->plan_host, plan.bytes, hipMemcpyHostToDevice, s));
  const uint8_t* pd = static_cast<const uint8_t*>(arena->plan_dev);
  plan.launch(pd, s);
  launch_copy_items(reinterpret_cast<const CopyItem*>(pd + off_copies), n, static_cast<uint8_t*>(arena->slabs_dev), s);
  HIP_TRY(hipMemcpyAsync(arena->slabs_host, arena->slabs_dev, slab_bytes, hipMemcpyDeviceToHost, s));
  for (uint32_t k = 0; k < n; ++k) jobs[which[k]]->readback = static_cast<uint8_t*>(arena->slabs_host) + copies[k].dst_offset;
  HIP_TRY(long_wait_stream(s));
  return DMI_OK;
}

// Record prep of a whole batch (after every job's tables were normalised on the host, in plan mode): the coding tables of all
// jobs travel in one copy and are scattered to their buffers by one kernel; then one launch per prep kernel.
static int run_phase_b_batched(dmi_job** jobs, const std::vector<uint32_t>& which, const std::vector<std::vector<KernelStep>>& steps, BatchArena* arena, hipStream_t s) {
  if (which.empty()) return DMI_OK;
  BatchPlan plan;
  plan.add(steps, kPrepLevels);
  std::vector<CopyItem> items;
  size_t table_bytes = 0;
  for (uint32_t j : which)
    for (const auto& p : jobs[j]->run.pending) { items.push_back(CopyItem{p.dst, 0, (uint64_t)p.bytes}); table_bytes += (p.bytes + 255) & ~(size_t)255; }
  const size_t off_items = plan.reserve(items.size() * sizeof(CopyItem));
  const size_t off_tables = plan.reserve(table_bytes);
  int rc;
  if ((rc = BatchArena::reserve_pair(arena->plan_host, arena->plan_dev, arena->plan_cap, plan.bytes))) return rc;
  uint8_t* ph = static_cast<uint8_t*>(arena->plan_host);
  plan.fill(ph);
  {
    size_t at = off_tables, k = 0;
    for (uint32_t j : which)
      for (const auto& p : jobs[j]->run.pending) {
        std::memcpy(ph + at, p.src, p.bytes);   // (sources are padded to whole 16-byte words by their owners)
        items[k++].dst_offset = at;
        at += (p.bytes + 255) & ~(size_t)255;
      }
  }
  std::memcpy(ph + off_items, items.data(), items.size() * sizeof(CopyItem));
  HIP_TRY(hipMemcpyAsync(arena->plan_dev, arena->plan_host, plan.bytes, hipMemcpyHostToDevice, s));
  const uint8_t* pd = static_cast<const uint8_t*>(arena->plan_dev);
  launch_scatter_items(reinterpret_cast<const CopyItem*>(pd + off_items), (uint32_t)items.size(), pd, s);
  plan.launch(pd, s);
  return DMI_OK;
}

// Device form of a batch: the phases, the table stage and the record prep of ALL jobs are planned together (one upload, one
// launch per (level, kernel)), the chain descriptors are written by k_tables, and the chains follow on the same stream — the
// host waits for the first time when everything has been coded.  Read-back: one packed arena (coded bytes + serialised tables),
// one table of {offset, length, error}, 128 scratch bytes per attribute.
// A batch is begun (plan, upload, launches: returns without waiting) and finished (wait, read back, splice) separately, so that
// two halves of a large batch can be in flight on two streams: the data-parallel kernels of the second half run under the
// chain launch of the first, which is latency-bound by its longest stream and leaves the vector units idle.
static int parallel_items(uint32_t n, uint32_t n_threads, int device, const std::function<int(uint32_t)>& fn) {
  n_threads = std::max(1u, std::min(n, n_threads));
  std::vector<int> rcs(n_threads, DMI_OK);
  std::vector<std::string> errs(n_threads);
  auto work = [&](uint32_t t) {
    if (hipSetDevice(device) != hipSuccess) { rcs[t] = DMI_ERR_HIP; errs[t] = "hipSetDevice"; return; }
    const uint32_t lo = (uint32_t)((uint64_t)n * t / n_threads), hi = (uint32_t)((uint64_t)n * (t + 1) / n_threads);
    for (uint32_t k = lo; k < hi; ++k) { const int rc = fn(k); if (rc) { rcs[t] = rc; errs[t] = g_last_error; return; } }
  };
  run_threads(n_threads, work);
  for (uint32_t t = 0; t < n_threads; ++t) if (rcs[t]) return fail(rcs[t], errs[t]);
  return DMI_OK;
}

struct DeviceBatch {
  std::vector<dmi_job*> jobs;
  std::vector<dmi_buffer*> outs;
  BatchArena* arena = nullptr;
  hipStream_t s = nullptr;
  uint32_t n_threads = 1;
  int device = 0;
  std::vector<uint32_t> first_desc, first_att;
  uint32_t n_streams = 0, n_atts = 0, n_descs = 0;
  size_t launches = 0;
  bool sparse_chains = false;
  double t_plan = 0, t_wait = 0, t_bytes = 0, t_splice = 0;
  ~DeviceBatch() {
    if (!arena) return;
    if (s) (void)long_wait_stream(s);   // (also on error paths: nothing of this batch may still be writing into the arena when the next one takes it)
    release_batch_arena(arena);
  }

  int begin() {
    const uint32_t n = (uint32_t)jobs.size();
    int rc;
    const auto t0 = std::chrono::steady_clock::now();
    first_desc.assign(n + 1, 0);
    first_att.assign(n + 1, 0);
    for (uint32_t j = 0; j < n; ++j) { first_desc[j + 1] = first_desc[j] + count_streams(jobs[j]); first_att[j + 1] = first_att[j] + (uint32_t)jobs[j]->atts.size(); }
    n_streams = first_desc[n]; n_atts = first_att[n]; n_descs = n_streams + n_atts;
    const size_t order_at = (size_t)n_descs * sizeof(ChainDesc), counter_at = order_at + (((size_t)n_streams * 4 + 15) & ~(size_t)15);
    if ((rc = arena->reserve_descs(counter_at + 16))) return rc;
    ChainDesc* descs_dev = static_cast<ChainDesc*>(arena->descs_dev);
    // ---- plan (host threads; no HIP call) ----
    std::vector<std::vector<KernelStep>> steps(n);
    if ((rc = parallel_items(n, n_threads, device, [&](uint32_t j) {
          dmi_job* job = jobs[j];
          job->readback = nullptr;
          set_step_sink(&steps[j]);
          int r = encode_phase_a(job, true);
          const size_t n_a = steps[j].size();
          if (!r) r = encode_phase_b_dev(job, descs_dev + first_desc[j], descs_dev + n_streams + first_att[j]);
          set_step_sink(nullptr);
          for (size_t k = n_a; k < steps[j].size(); ++k) steps[j][k].level += kStepLevels;
          return r;
        }))) return rc;
    BatchPlan plan;
    plan.add(steps, kStepLevels + kPrepLevels);
    launches = plan.groups.size();
    // stream order for the chain kernel (longest first), scratch-word copies, capacities
    std::vector<uint64_t> length(n_streams);
    size_t cap_sum = 0;
    for (uint32_t j = 0; j < n; ++j)
      for (size_t k = 0; k < jobs[j]->run.descs.size(); ++k) { const ChainDesc& d = jobs[j]->run.descs[k]; length[first_desc[j] + k] = d.n; cap_sum += ((size_t)d.cap + 31) & ~(size_t)15; }
    for (uint32_t j = 0; j < n; ++j) for (auto& a : jobs[j]->atts) cap_sum += ((size_t)a.hdr_cap + 31) & ~(size_t)15;
    std::vector<uint32_t> by_length(n_streams);
    for (uint32_t k = 0; k < n_streams; ++k) by_length[k] = k;
    std::stable_sort(by_length.begin(), by_length.end(), [&](uint32_t x, uint32_t y) { return length[x] > length[y]; });
    std::vector<CopyItem> copies;
    copies.reserve(n_atts);
    for (uint32_t j = 0; j < n; ++j) for (auto& a : jobs[j]->atts) copies.push_back(CopyItem{a.small.p, (uint64_t)copies.size() * 128u, 128u});
    const size_t off_copies = plan.reserve(copies.size() * sizeof(CopyItem));
    const size_t off_order = plan.reserve((size_t)n_streams * 4);
    if ((rc = BatchArena::reserve_pair(arena->plan_host, arena->plan_dev, arena->plan_cap, plan.bytes))) return rc;
    if ((rc = BatchArena::reserve_pair(arena->slabs_host, arena->slabs_dev, arena->slabs_cap, (size_t)n_atts * 128))) return rc;
    if ((rc = arena->reserve(cap_sum, (size_t)(n_descs + 1) * sizeof(PackEntry)))) return rc;
    uint8_t* ph = static_cast<uint8_t*>(arena->plan_host);
    plan.fill(ph);
    std::memcpy(ph + off_copies, copies.data(), copies.size() * sizeof(CopyItem));
    std::memcpy(ph + off_order, by_length.data(), (size_t)n_streams * 4);
    // ---- the whole encode: one upload, then launches only ----
    HIP_TRY(hipMemcpyAsync(arena->plan_dev, arena->plan_host, plan.bytes, hipMemcpyHostToDevice, s));
    const uint8_t* pd = static_cast<const uint8_t*>(arena->plan_dev);
    plan.launch(pd, s);
    {
      uint64_t longest = 0, total = 0;
      for (uint64_t v : length) { longest = std::max(longest, v); total += v; }
      sparse_chains = chain_launch_sparse(longest, total, n_streams);
      launch_chains(descs_dev, reinterpret_cast<const uint32_t*>(pd + off_order), n_streams, reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(arena->descs_dev) + counter_at), sparse_chains, s);
    }
    launch_pack_streams(descs_dev, n_descs, static_cast<PackEntry*>(arena->table_dev), static_cast<uint8_t*>(arena->bytes_dev), s);
    launch_copy_items(reinterpret_cast<const CopyItem*>(pd + off_copies), n_atts, static_cast<uint8_t*>(arena->slabs_dev), s);
    HIP_TRY(hipMemcpyAsync(arena->table_host, arena->table_dev, (size_t)(n_descs + 1) * sizeof(PackEntry), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(arena->slabs_host, arena->slabs_dev, (size_t)n_atts * 128, hipMemcpyDeviceToHost, s));
    t_plan = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return DMI_OK;
  }

  int finish() {
    const uint32_t n = (uint32_t)jobs.size();
    int rc;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t1 = now();
    HIP_TRY(long_wait_stream(s)   /* the chain launch: bounded by its longest stream, milliseconds */);
    const auto t2 = now();
    const PackEntry* table = static_cast<const PackEntry*>(arena->table_host);
    const size_t total = (size_t)table[n_descs].offset;
    if ((rc = arena->reserve_host(total))) return rc;
    if (total) HIP_TRY(hipMemcpyAsync(arena->bytes_host, arena->bytes_dev, total, hipMemcpyDeviceToHost, s));
    HIP_TRY(long_wait_stream(s));
    const auto t3 = now();
    const uint8_t* bytes_host = static_cast<const uint8_t*>(arena->bytes_host);
    if ((rc = parallel_items(n, n_threads, device, [&](uint32_t j) {
          dmi_job* job = jobs[j];
          const uint32_t na = (uint32_t)job->atts.size();
          job->readback = static_cast<uint8_t*>(arena->slabs_host) + (size_t)first_att[j] * 128;
          job->run.pin_off.assign(na, 0);
          job->run.hdr_ptr.assign(na, nullptr);
          job->run.hdr_len.assign(na, 0);
          for (uint32_t i = 0; i < na; ++i) {
            job->run.pin_off[i] = (size_t)i * 128;
            const uint32_t* small = reinterpret_cast<const uint32_t*>(job->readback + job->run.pin_off[i]);
            int frc = check_device_flags(small, i);
            if (!frc) frc = check_value_bounds(job->atts[i], small, i);
            if (frc) return frc;
            job->run.aux[i].zero_prob = (uint8_t)small[14];
            job->run.aux[i].count = small[15];
            const PackEntry& h = table[n_streams + first_att[j] + i];
            job->run.hdr_ptr[i] = bytes_host + h.offset;
            job->run.hdr_len[i] = h.len;
          }
          int r = encode_phase_c_packed(job, table, first_desc[j], bytes_host);
          if (!r) r = encode_phase_c3(job, outs[j]);
          return r;
        }))) return rc;
    t_wait = ms(t1, t2); t_bytes = ms(t2, t3); t_splice = ms(t3, now());
    return DMI_OK;
  }

  void trace(const char* name) const {   // per-stream chain clocks (100 MHz ticks written by the emitters) + host stages
    double sum_ms = 0, max_ms = 0, steps = 0, big_steps = 0, big_ms = 0;
    for (dmi_job* job : jobs)
      for (uint32_t i = 0; i < (uint32_t)job->atts.size(); ++i) {
        const uint32_t* small = reinterpret_cast<const uint32_t*>(job->readback + job->run.pin_off[i]);
        const double r = small[12] * 1e-5, x = job->run.aux[i].desc >= 0 ? small[13] * 1e-5 : 0.0;
        const double ns = (double)job->atts[i].n_sym, nx = job->run.aux[i].desc >= 0 ? (double)job->run.aux[i].count : 0.0;
        sum_ms += r + x; max_ms = std::max({max_ms, r, x}); steps += ns + nx;
        if (ns > 50000) { big_steps += ns; big_ms += r; }
      }
    std::fprintf(stderr, "[dmi] %s: %zu jobs, %u streams (%s chain launch), %zu launches; plan + issue %.2f ms, wait for the stream %.2f, byte read-back %.2f, splice %.2f; chains: %.0f steps, stream times sum %.1f ms "
                 "(/1024 walkers = %.2f), longest %.2f ms, %.1f ns/step (%.1f on rANS streams > 50k symbols)\n", name, jobs.size(), n_streams, sparse_chains ? "sparse" : "dense", launches, t_plan, t_wait, t_bytes, t_splice, steps, sum_ms,
                 sum_ms / 1024.0, max_ms, sum_ms * 1e6 / std::max(1.0, steps), big_ms * 1e6 / std::max(1.0, big_steps));
  }
};

// The two streams of a split batch (process lifetime, one pair per device; created back to back so that they land on different
// hardware queues — two streams that share a queue run their kernels strictly one after the other).
static bool batch_stream_pair(int device, hipStream_t& a, hipStream_t& b) {
  struct Pair { int device; hipStream_t a, b; };
  static std::mutex m;
  static std::vector<Pair> pairs;
  std::lock_guard<std::mutex> lock(m);
  for (auto& e : pairs) if (e.device == device) { a = e.a; b = e.b; return true; }
  Pair p{device, nullptr, nullptr};
  if (hipStreamCreateWithFlags(&p.a, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&p.b, hipStreamNonBlocking) != hipSuccess) return false;
  pairs.push_back(p);
  a = p.a; b = p.b;
  return true;
}

static int jobs_encode_device(dmi_job** jobs, uint32_t n, dmi_buffer* outs, uint32_t n_threads, bool trace) {
  const int device = jobs[0]->cfg.device;
  // Large batches run as two halves in flight: the jobs with the longest streams first (≈ 45 % of the symbols), the rest behind
  // them on a second stream — its data-parallel kernels (and the host's planning of it) run under the first half's chain launch.
  std::vector<uint32_t> order(n);
  for (uint32_t j = 0; j < n; ++j) order[j] = j;
  auto symbols = [&](uint32_t j) { uint64_t t = 0; for (auto& a : jobs[j]->atts) t += a.n_sym; return t; };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return symbols(x) > symbols(y); });
  uint64_t total = 0;
  for (uint32_t j = 0; j < n; ++j) total += symbols(j);
  uint32_t n_first = n;
  hipStream_t main_stream = jobs[0]->stream, side = nullptr;
  bool library_streams = true;   // a caller's stream (dmi_config.stream) is honoured: everything stays on it
  for (uint32_t j = 0; j < n; ++j) if (jobs[j]->cfg.stream) library_streams = false;
  if (n >= 32 && library_streams && dbg().split && batch_stream_pair(device, main_stream, side)) {
    uint64_t acc = 0;
    n_first = 0;
    while (n_first < n && acc * 100 < total * 45) acc += symbols(order[n_first++]);
    if (n_first < 8 || n - n_first < 8) n_first = n;
  }
  // (The chain launch is bounded by its LONGEST stream — one walker steps at ≈ 17 ns: a 300 K-symbol position stream is 4.8 ms while the sum of all
  // streams over 1024 walkers is 0.5 ms.  Taking the few longest jobs out of the batch and coding them in the hybrid form on host threads was tried
  // in round 4 — same bytes, 7.8 → 9.1 ms per 256-mesh encode — and removed: scripts/experiments/README.md.)
  DeviceBatch first, second;
  auto fill = [&](DeviceBatch& b, uint32_t lo, uint32_t hi, hipStream_t s) {
    for (uint32_t k = lo; k < hi; ++k) { b.jobs.push_back(jobs[order[k]]); b.outs.push_back(&outs[order[k]]); }
    b.arena = acquire_batch_arena(device);
    b.s = s; b.n_threads = n_threads; b.device = device;
  };
  fill(first, 0, n_first, main_stream);
  int rc;
  // At most two batch encodes per DEVICE have their chain launch in flight (round 6).  k_chains is a persistent launch of walker / emitter wavefront pairs
  // placed per SIMD: a transcoder's two encode threads are what the device takes side by side — with two transcoders on one device (dmi_transcode_assets
  // over [0, 0], two callers sharing a GPU) four launches held each other's slots and a 7 ms encode took 70–150 ms (scripts/experiments/one_process_weak.py).
  ChainSlot chain_slot(device);
  if ((rc = first.begin())) return rc;
  if (n_first < n) {
    fill(second, n_first, n, side);
    if ((rc = second.begin())) { (void)hipStreamSynchronize(first.s); (void)hipStreamSynchronize(second.s); return rc; }
  }
  rc = first.finish();
  if (n_first < n) {
    if (rc) (void)hipStreamSynchronize(second.s);   // a failed first half: let the second drain, report the first error
    else rc = second.finish();
  }
  if (rc) return rc;
  if (trace) { first.trace(n_first < n ? "batch, device form, first half" : "batch, device form"); if (n_first < n) second.trace("batch, device form, second half"); }
  return DMI_OK;
}

static int jobs_encode_impl(dmi_job** jobs, uint32_t n, dmi_buffer* outs);
int dmi_jobs_encode(dmi_job** jobs, uint32_t n, dmi_buffer* outs) {
  DebugScope debug_scope((jobs && n && jobs[0]) ? &jobs[0]->debug : nullptr);
  if (!jobs || !outs || n == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  for (uint32_t j = 0; j < n; ++j) outs[j] = dmi_buffer{};
  const int rc = jobs_encode_impl(jobs, n, outs);
  if (rc) {   // all or nothing: no output of a failed batch is left allocated (the error text survives the frees)
    const std::string why = g_last_error;
    dmi_free_many(outs, n);
    g_last_error = why;
  }
  return rc;
}
// One process, several GPUs: the jobs are grouped by the device they live on and every group is coded by its own dmi_jobs_encode
// on its own host thread — the devices run concurrently, the call returns when all have finished.  All or nothing.
int dmi_jobs_encode_devices(dmi_job** jobs, uint32_t n, dmi_buffer* outs) {
  DebugScope debug_scope((jobs && n && jobs[0]) ? &jobs[0]->debug : nullptr);
  if (!jobs || !outs || n == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  for (uint32_t j = 0; j < n; ++j) { outs[j] = dmi_buffer{}; if (!jobs[j]) return fail(DMI_ERR_INVALID_ARGUMENT, "null job"); }
  std::vector<int> devices;
  for (uint32_t j = 0; j < n; ++j) if (std::find(devices.begin(), devices.end(), jobs[j]->cfg.device) == devices.end()) devices.push_back(jobs[j]->cfg.device);
  if (devices.size() == 1) return dmi_jobs_encode(jobs, n, outs);
  std::vector<int> rcs(devices.size(), DMI_OK);
  std::vector<std::string> errs(devices.size());
  auto work = [&](size_t g) {
    std::vector<dmi_job*> mine;
    std::vector<uint32_t> at;
    for (uint32_t j = 0; j < n; ++j) if (jobs[j]->cfg.device == devices[g]) { mine.push_back(jobs[j]); at.push_back(j); }
    std::vector<dmi_buffer> got(mine.size());
    rcs[g] = dmi_jobs_encode(mine.data(), (uint32_t)mine.size(), got.data());
    if (rcs[g]) { errs[g] = g_last_error; return; }
    for (size_t k = 0; k < at.size(); ++k) outs[at[k]] = got[k];
  };
  std::vector<dmi::Thread> th;
  for (size_t g = 1; g < devices.size(); ++g) th.emplace_back(with_debug(work), g);
  work(0);
  for (auto& x : th) x.join();
  for (size_t g = 0; g < devices.size(); ++g)
    if (rcs[g]) {
      dmi_free_many(outs, n);
      return fail(rcs[g], "device " + std::to_string(devices[g]) + ": " + errs[g]);
    }
  return DMI_OK;
}
static int jobs_encode_impl(dmi_job** jobs, uint32_t n, dmi_buffer* outs) {
  for (uint32_t j = 0; j < n; ++j) if (!jobs[j] || jobs[j]->cfg.device != jobs[0]->cfg.device) return fail(DMI_ERR_INVALID_ARGUMENT, "batched jobs must live on one device");
  const int device = jobs[0]->cfg.device;
  // Small meshes are launch-bound, so whatever stays per job (table normalisation; the phases of jobs that cannot be planned ahead) runs on several host
  // threads, each walking a contiguous slice of the jobs (jobs that own their stream then also overlap on the GPU).
  const uint32_t thread_cap = dbg().batch_threads ? dbg().batch_threads : 16u;
  const uint32_t n_threads = std::max(1u, std::min({n, (uint32_t)host_threads(), std::max(1u, thread_cap)}));
  auto parallel = [&](auto&& fn, bool sync_after = true) -> int {
    std::vector<int> rcs(n_threads, DMI_OK);
    std::vector<std::string> errs(n_threads);
    auto work = [&](uint32_t t) {
      if (hipSetDevice(device) != hipSuccess) { rcs[t] = DMI_ERR_HIP; errs[t] = "hipSetDevice"; return; }
      const uint32_t lo = (uint32_t)((uint64_t)n * t / n_threads), hi = (uint32_t)((uint64_t)n * (t + 1) / n_threads);
      for (uint32_t j = lo; j < hi; ++j) { const int rc = fn(j); if (rc) { rcs[t] = rc; errs[t] = g_last_error; return; } }
      if (sync_after) {   // each worker waits for its own slice's streams
        hipStream_t last = nullptr;
        for (uint32_t j = lo; j < hi; ++j) {
          if (j > lo && jobs[j]->stream == last) continue;
          last = jobs[j]->stream;
          if (hipStreamSynchronize(last) != hipSuccess) { rcs[t] = DMI_ERR_HIP; errs[t] = "hipStreamSynchronize"; return; }
        }
      }
    };
    run_threads(n_threads, work);
    for (uint32_t t = 0; t < n_threads; ++t) if (rcs[t]) return fail(rcs[t], errs[t]);
    return DMI_OK;
  };
  hipStream_t s = jobs[0]->stream;
  int rc;
  const bool trace = dbg_on(DMI_DBG_TRACE);
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  const auto t0 = now();
  // Pipeline per worker: phase A of every job of its share is queued first; then, job by job, the worker waits for
  // that job's histograms, normalises its tables and queues the record prep — host work of early jobs overlaps the
  // data-parallel kernels of later ones.  The chains of ALL jobs then run in one launch: a long-running kernel per job
  // would pin one of the few hardware queues each and serialise the batch (measured: 114 ms instead of 8).
  BatchArena* arena = acquire_batch_arena(device);
  struct Release { BatchArena* a; ~Release() { if (a) release_batch_arena(a); } } release{arena};
  // jobs whose phase A can be planned ahead (no mid-phase host wait, no per-job event timing) share one launch per kernel
  std::vector<uint32_t> batched;
  std::vector<uint8_t> is_batched(n, 0);
  for (uint32_t j = 0; j < n; ++j) {
    jobs[j]->readback = nullptr;
    bool ok = !jobs[j]->have_events && !dbg_on(DMI_DBG_NO_BATCHED_PHASES);
    for (auto& a : jobs[j]->atts) if (a.port == kToBits) ok = false;
    if (ok) { batched.push_back(j); is_batched[j] = 1; }
  }
  HIP_TRY(hipSetDevice(device));
  {
    bool all_device = batched.size() == n;
    for (uint32_t j = 0; j < n && all_device; ++j) all_device = jobs[j]->dev_tables;
    if (all_device) { release.a = nullptr; release_batch_arena(arena); return jobs_encode_device(jobs, n, outs, n_threads, trace); }
  }
  if ((rc = run_phase_a_batched(jobs, batched, arena, s))) return rc;
  const auto t1 = now();
  std::vector<std::vector<KernelStep>> b_steps(n);   // record-prep steps of the batched jobs (filled by the workers)
  std::vector<uint32_t> order(n);
  for (uint32_t j = 0; j < n; ++j) order[j] = j;
  auto job_size = [&](uint32_t j) { uint64_t t = 0; for (auto& a : jobs[j]->atts) t += a.n_sym; return t; };
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return job_size(x) > job_size(y); });
  {
    std::vector<int> rcs(n_threads, DMI_OK);
    std::vector<std::string> errs(n_threads);
    auto work = [&](uint32_t t) {
      auto bail = [&](int rc_, const std::string& e) { rcs[t] = rc_; errs[t] = e; };
      if (hipSetDevice(device) != hipSuccess) return bail(DMI_ERR_HIP, "hipSetDevice");
      // jobs dealt round-robin in descending size: every worker gets a mix, its first job is one of the largest
      std::vector<uint32_t> mine;
      for (uint32_t k = t; k < n; k += n_threads) mine.push_back(order[k]);
      const auto w0 = std::chrono::steady_clock::now();
      for (uint32_t j : mine) { if (is_batched[j]) continue; const int r = run_phase_a(jobs[j]); if (r) return bail(r, g_last_error); }
      const auto w1 = std::chrono::steady_clock::now();
      double wait_ms = 0, b_ms = 0;
      for (uint32_t j : mine) {
        dmi_job* job = jobs[j];
        const auto x0 = std::chrono::steady_clock::now();
        if (!is_batched[j] && hipStreamSynchronize(job->stream) != hipSuccess) return bail(DMI_ERR_HIP, "hipStreamSynchronize");
        const auto x1 = std::chrono::steady_clock::now();
        int r;
        if (is_batched[j]) { set_step_sink(&b_steps[j]); r = encode_phase_b(job, true); set_step_sink(nullptr); }
        else r = encode_phase_b(job);
        if (r) return bail(r, g_last_error);
        const auto x2 = std::chrono::steady_clock::now();
        wait_ms += std::chrono::duration<double, std::milli>(x1 - x0).count();
        b_ms += std::chrono::duration<double, std::milli>(x2 - x1).count();
      }
      const auto w2 = std::chrono::steady_clock::now();
      for (uint32_t j : mine) if (hipStreamSynchronize(jobs[j]->stream) != hipSuccess) return bail(DMI_ERR_HIP, "hipStreamSynchronize");
      if (trace && t == 0) std::fprintf(stderr, "[dmi] worker 0: %zu jobs, phase A issue %.2f ms, waits for histograms %.2f, phase B host+issue %.2f, final wait %.2f\n", mine.size(),
                                        std::chrono::duration<double, std::milli>(w1 - w0).count(), wait_ms, b_ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w2).count());
    };
    run_threads(n_threads, work);
    for (uint32_t t = 0; t < n_threads; ++t) if (rcs[t]) return fail(rcs[t], errs[t]);
  }
  const auto t3 = now();
  {
    std::vector<std::vector<KernelStep>> only;
    only.reserve(batched.size());
    for (uint32_t j : batched) only.push_back(std::move(b_steps[j]));
    if ((rc = run_phase_b_batched(jobs, batched, only, arena, s))) return rc;
  }
  const auto t4 = now();
  std::vector<ChainDesc> all;
  for (uint32_t j = 0; j < n; ++j) all.insert(all.end(), jobs[j]->run.descs.begin(), jobs[j]->run.descs.end());
  // the chain kernel serves the streams longest first (its pairs pull work; see k_chains)
  std::vector<uint32_t> by_length(all.size());
  for (uint32_t k = 0; k < (uint32_t)all.size(); ++k) by_length[k] = k;
  std::stable_sort(by_length.begin(), by_length.end(), [&](uint32_t x, uint32_t y) { return all[x].n > all[y].n; });
  DevMem descs_dev;   // descriptors | order | pull counter
  const size_t order_at = all.size() * sizeof(ChainDesc), counter_at = order_at + ((all.size() * sizeof(uint32_t) + 15) & ~(size_t)15);
  if ((rc = descs_dev.alloc(counter_at + 16))) return rc;
  HIP_TRY(hipMemcpyAsync(descs_dev.p, all.data(), order_at, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(static_cast<uint8_t*>(descs_dev.p) + order_at, by_length.data(), all.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
  {
    uint64_t longest = 0, total = 0;
    for (const ChainDesc& cd : all) { longest = std::max<uint64_t>(longest, cd.n); total += cd.n; }
    launch_chains(descs_dev.as<ChainDesc>(), reinterpret_cast<const uint32_t*>(static_cast<uint8_t*>(descs_dev.p) + order_at), (uint32_t)all.size(),
                  reinterpret_cast<uint32_t*>(static_cast<uint8_t*>(descs_dev.p) + counter_at), chain_launch_sparse(longest, total, (uint32_t)all.size()), s);
  }
  HIP_TRY(long_wait_stream(s));
  const auto t5 = now();
  // read-back: every stream of every job packed into one arena on the device → one table copy + one byte copy
  const uint32_t n_streams = (uint32_t)all.size();
  std::vector<uint32_t> first_desc(n, 0);
  size_t cap_sum = 0;
  {
    uint32_t at = 0;
    for (uint32_t j = 0; j < n; ++j) { first_desc[j] = at; at += (uint32_t)jobs[j]->run.descs.size(); }
    for (const ChainDesc& d : all) cap_sum += ((size_t)d.cap + 31) & ~(size_t)15;
  }
  if ((rc = arena->reserve(cap_sum, (size_t)(n_streams + 1) * sizeof(PackEntry)))) return rc;
  launch_pack_streams(descs_dev.as<ChainDesc>(), n_streams, static_cast<PackEntry*>(arena->table_dev), static_cast<uint8_t*>(arena->bytes_dev), s);
  HIP_TRY(hipMemcpyAsync(arena->table_host, arena->table_dev, (size_t)(n_streams + 1) * sizeof(PackEntry), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const PackEntry* table = static_cast<const PackEntry*>(arena->table_host);
  const size_t total = (size_t)table[n_streams].offset;
  if ((rc = arena->reserve_host(total))) return rc;
  if (total) HIP_TRY(hipMemcpyAsync(arena->bytes_host, arena->bytes_dev, total, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  for (uint32_t j = 0; j < n; ++j) if ((rc = encode_phase_c_packed(jobs[j], table, first_desc[j], static_cast<const uint8_t*>(arena->bytes_host)))) return rc;
  const auto t6 = now();
  if ((rc = parallel([&](uint32_t j) { return encode_phase_c3(jobs[j], &outs[j]); }, false))) return rc;
  if (trace) std::fprintf(stderr, "[dmi] batch of %u (%zu with batched phases) on %u host threads: data-parallel phases %.2f ms, tables (host threads) %.2f + record prep plan/upload/launch %.2f, chains (%zu streams, one launch) %.2f, packed read-back %.2f, splice %.2f\n", n, batched.size(), n_threads, ms(t0, t1), ms(t1, t3), ms(t3, t4), all.size(), ms(t4, t5), ms(t5, t6), ms(t6, now()));
  return DMI_OK;
}

int dmi_encode_attributes_batch(const dmi_batch_item* items, uint32_t n, const dmi_config* cfg_in, dmi_buffer* outs) {
  DebugScope debug_scope(cfg_in ? cfg_in->debug : nullptr);
  if (!items || !outs || n == 0) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  dmi_config cfg{};
  if (cfg_in) cfg = *cfg_in;
  HIP_TRY(hipSetDevice(cfg.device));
  hipStream_t own = nullptr;
  if (!cfg.stream) { HIP_TRY(hipStreamCreate(&own)); cfg.stream = own; }
  std::vector<dmi_job*> jobs(n, nullptr);
  int rc = DMI_OK;
  for (uint32_t j = 0; j < n && !rc; ++j) rc = dmi_job_create(items[j].atts, items[j].tables, items[j].n_atts, items[j].seeds, items[j].n_seeds, &cfg, &jobs[j]);
  if (!rc) rc = dmi_jobs_encode(jobs.data(), n, outs);
  for (auto* j : jobs) dmi_job_destroy(j);
  if (own) (void)hipStreamDestroy(own);
  return rc;
}
