// dmi_encode.cpp — the encode pipeline of ONE resident job: value ranges → coding-order gather + quantize → predict + transform →
// histograms → table stage (k_tables; host form behind DMI_HOST_TABLES) → { record prep → walker/emitter rANS/rABS chains | symbols read
// back → one host core per stream (hybrid form) } → byte splice.  The phases are split at their host synchronisation points so that the
// batch drivers (dmi_batch.cpp) can run the same phase of many jobs in one launch per kernel.
#include "dmi_job.hpp"

using namespace dmi;

int encode_phase_a(dmi_job* job, bool plan_only) {   // device: ranges → coding-order portabilization → predict → histograms; async read-back
  // plan_only: the caller has set a step sink — every launch below is collected, not issued, and the read-back is the caller's
  if (!plan_only) HIP_TRY(hipSetDevice(job->cfg.device));
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  if (!plan_only && !job->pinned) HIP_TRY(hipHostMalloc(&job->pinned, job->pinned_bytes, hipHostMallocDefault));
  uint8_t* pinned = static_cast<uint8_t*>(job->pinned);
  const bool timed = job->have_events;
  // ---- stage 1: value ranges (streamed over the unique values) ---------------------------------------------
  const bool early = job->early != nullptr && !plan_only;   // ranges + quantization ran before the host walks (EarlyQuant): the pass gathers packed values
  if (early) HIP_TRY(hipStreamWaitEvent(s, job->early->t1, 0));
  if (timed) HIP_TRY(hipEventRecord(job->ev[0], s));
  if (!early) {
    RangeArgs ra{};
    for (auto& a : job->atts) {
      RangeAtt& r = ra.a[ra.count++];
      r.raw = a.raw.as<float>();
      r.partials = a.partials.as<float>();
      r.meta = a.meta.as<float>();
      r.small = a.small.as<uint32_t>();   // zeroed here; [0..1] := {INT_MAX, INT_MIN}; [4] := zero-length normal seen
      r.zero = a.meta.as<uint32_t>();   // meta, histogram, summaries: contiguous in the slab slot
      r.zero_words = (a.meta.bytes + a.hist.bytes + a.summary.bytes) / 4;
      r.n = a.desc.num_unique;
      r.N = a.desc.num_components;
      r.kind = a.port == kCoordwise ? 0 : (a.port == kOct ? 1 : 2);
      if (ra.count == kMaxRangeAtts) { launch_value_ranges(ra, s); ra.count = 0; }
    }
    launch_value_ranges(ra, s);
  }
  // ---- stage 2: portabilization in coding order + predict + transform ---------------------------------------
  if (timed) HIP_TRY(hipEventRecord(job->ev[1], s));
  // small: [0..1] minmax, [2..3] counters, [4] zero-normal flag, [5] hist overflow, [8..13] coder lengths / flags / ticks
  for (size_t ti = 0; ti < job->tables.size(); ++ti) {
    TableDev& t = job->tables[ti];
    if (t.alias_of >= 0) continue;
    if (early) {   // (every attribute of an adopted early stage is per-point and sits on the position's table)
      if ((size_t)job->atts[0].table != ti) continue;
      GatherRecArgs ga{};
      ga.s2p = t.s2p.as<uint32_t>(); ga.n = t.n_seq; ga.rec = job->early->rec;
      for (size_t i = 0; i < job->atts.size(); ++i) {
        AttJob& a = job->atts[i];
        if (a.qfmt == QF_P64) ga.qs_pos = a.qs.as<uint64_t>();
        else if (a.qfmt == QF_B16) ga.qs_nrm = a.qs.as<uint16_t>();
        else ga.qs_uv = a.qs.as<uint32_t>();
      }
      // the ranges, the seeded scratch words, the zero-normal flag and the joint min/max (folded from the stage's per-block pairs) into the job's slab
      // slots ([small 64 B][meta 64 B]): the gather's first block
      ga.slots = job->early->slots_for([&](size_t i) { return job->atts[i].small.as<uint32_t>(); });
      launch_seq_gather_rec(ga, s);
      continue;
    }
    QuantArgs qa{};
    auto flush = [&]() {
      if (qa.count) launch_seq_quantize(t.s2p_sorted.p ? t.s2p_sorted.as<uint32_t>() : t.s2p.as<uint32_t>(), t.s2p_sorted.p ? t.sorted_dest.as<uint32_t>() : nullptr, t.n_seq, qa, s);
      qa.count = 0;
    };
    for (auto& a : job->atts) {
      if ((size_t)a.table != ti) continue;
      QuantAtt& g = qa.a[qa.count++];
      g.raw = a.raw.as<float>();
      g.s2v = a.s2v.as<uint32_t>();
      g.qs = a.qs.p;
      g.fmt = a.qfmt;
      g.ipartials = a.ipartials.as<int32_t>();
      g.meta = a.meta.as<float>();
      g.maxq = (float)(uint64_t)((1ull << a.bits) - 1ull);
      g.kind = a.port == kCoordwise ? 0 : (a.port == kOct ? 1 : 2);
      g.N = a.desc.num_components;
      if (qa.count == kMaxGather) flush();
    }
    flush();
  }
  if (!early) {   // (an early stage left the min/max in the slot words copied above)
    MinMaxArgs ma{};
    for (auto& a : job->atts) {
      MinMaxAtt& m = ma.a[ma.count++];
      m.ipartials = a.ipartials.as<int32_t>();
      m.minmax = a.small.as<int32_t>();
      m.blocks = seq_quantize_blocks(job->tables[a.table].n_seq);
      if (ma.count == kMaxRangeAtts) { launch_i32_minmax_final(ma, s); ma.count = 0; }
    }
    launch_i32_minmax_final(ma, s);
  }
  OrientArgs fused_orient{};
  for (auto& a : job->atts) {
    const TableDev& t = job->tables[a.table];
    const int32_t* minmax = a.small.as<int32_t>();
    uint32_t* counters = a.small.as<uint32_t>() + 2;
    const uint32_t n = t.n_seq;
    if (n == 0) continue;
    if (a.fused_into >= 0) continue;   // predicted by its parent's fused sweep
    if (a.fused_nrm >= 0 || a.fused_uv >= 0) {
      FusedArgs fa{};
      fa.seq = t.seq.as<uint32_t>(); fa.c2r = t.c2r.as<uint32_t>(); fa.opp = t.opp.as<uint32_t>(); fa.n = n;
      if (t.frec.p) { fa.c2r = t.frec.as<uint32_t>(); fa.opp = t.frec.as<uint32_t>() + 4; fa.face_stride = 8u; }   // (face records: see TableDev::frec)
      fa.qs_pos = a.qs.p; fa.mm_pos = minmax; fa.sym_pos = a.sym.p;
      fa.packed = a.qfmt == QF_P64 ? 1u : 0u;
      fa.sym16 = a.sym16 ? 1u : 0u;
      fa.fan_hdr = t.fan_hdr.as<uint32_t>(); fa.fan_apex = t.fan_apex.as<uint32_t>(); fa.fan = t.fan.as<uint32_t>();
      if (a.fused_nrm >= 0) {
        AttJob& q = job->atts[a.fused_nrm];
        fa.qs_nrm = q.qs.p; fa.sym_nrm = q.sym.p; fa.flips = q.aux.as<uint8_t>(); fa.counters = q.small.as<uint32_t>() + 2;
        fa.flip_partials = q.flip_partials.as<uint32_t>(); q.flip_blocks = predict_fused_blocks(n);
        if (q.sym16) fa.sym16 |= 2u;
      }
      if (a.fused_uv >= 0) {
        AttJob& q = job->atts[a.fused_uv];
        fa.qs_uv = q.qs.p; fa.mm_uv = q.small.as<int32_t>(); fa.sym_uv = q.sym.p; fa.orient = q.aux.as<uint8_t>();
        fa.fix_list = q.fix_list.as<uint32_t>(); fa.fix_count = q.small.as<uint32_t>() + 3;
        if (q.sym16) fa.sym16 |= 4u;
      }
      launch_predict_fused(fa, s);
      if (a.fused_uv >= 0) { AttJob& q = job->atts[a.fused_uv]; fused_orient = OrientArgs{q.aux.as<uint8_t>(), q.summary.as<uint32_t>(), n, 0u}; }   // summarised by the histogram launch
      continue;
    }
    switch (a.scheme) {
      case kParallelogram:
        launch_pred_parallelogram_wrapped(t.seq.as<uint32_t>(), n, t.c2r.as<uint32_t>(), t.opp.as<uint32_t>(), a.qs.as<int32_t>(), minmax, a.nq, a.sym.p, a.sym16, s);
        break;
      case kDelta:
        launch_pred_delta_difference(n, a.qs.as<int32_t>(), a.nq, a.sym.p, a.sym16, s);
        break;
      case kNormal: {
        // the fan-row sweep with only the normal attribute: fans of this (attribute) table, positions through the parent's table
        const AttJob& p = job->atts[a.parent];
        FusedArgs fa{};
        fa.seq = t.seq.as<uint32_t>(); fa.c2r = job->tables[p.table].c2r.as<uint32_t>(); fa.opp = t.opp.as<uint32_t>(); fa.n = n;
        fa.qs_pos = p.qs.p; fa.packed = p.qfmt == QF_P64 ? 1u : 0u;   // (the parent's positions may be packed by its own fused sweep; this attribute's values are not)
        fa.qs_nrm = a.qs.p; fa.sym_nrm = a.sym.p; fa.flips = a.aux.as<uint8_t>(); fa.counters = counters;
        fa.flip_partials = a.flip_partials.as<uint32_t>(); a.flip_blocks = predict_fused_blocks(n);
        fa.sym16 = a.sym16 ? 2u : 0u;
        fa.fan_hdr = a.fan_hdr.as<uint32_t>(); fa.fan_apex = a.fan_apex.as<uint32_t>(); fa.fan = a.fan.as<uint32_t>();
        launch_predict_fused(fa, s);
        break;
      }
      case kTexCoord: {
        const AttJob& p = job->atts[a.parent];
        launch_pred_texcoord_wrapped(t.seq.as<uint32_t>(), n, t.c2r.as<uint32_t>(), a.qs.as<int32_t>(), job->tables[p.table].c2r.as<uint32_t>(), p.qs.p, p.qfmt, minmax, a.sym.p, a.sym16, a.aux.as<uint8_t>(), s);
        launch_orient_summary(a.aux.as<uint8_t>(), n, a.summary.as<uint32_t>(), nullptr, s);
        break;
      }
    }
  }
  // ---- stage 3: histograms ---------------------------------------------------------------------------
  if (timed) HIP_TRY(hipEventRecord(job->ev[2], s));
  std::vector<size_t>& pin_off = job->run.pin_off;
  pin_off.assign(n_atts, 0);
  HistArgs ha{};
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    if (a.port == kToBits) {
      // alphabet bound needs the value range: read min/max first
      int32_t mm[2];
      HIP_TRY(hipMemcpyAsync(mm, a.small.p, 8, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      const uint64_t span = (mm[1] >= mm[0]) ? (uint64_t)((int64_t)mm[1] - (int64_t)mm[0]) : 0;
      const uint64_t need = (a.transform == kWrapped ? span + 3 : 2 * span + 2);
      if (need > a.bins_cap) return fail(DMI_ERR_ALPHABET_TOO_LARGE, "custom attribute value range needs more than 2^20 symbols");
      a.bins = (uint32_t)need;
    }
    HistAtt& h = ha.a[ha.count++];
    h.sym = a.sym.p; h.sym16 = a.sym16 ? 1u : 0u; h.n = a.n_sym; h.hist = a.hist.as<uint32_t>(); h.bins = a.bins; h.overflow = a.small.as<uint32_t>() + 5;
    if (a.scheme == kNormal && a.flip_partials.p && a.n_sym) { h.flip_partials = a.flip_partials.as<uint32_t>(); h.flip_count = a.small.as<uint32_t>() + 2; h.n_flip_partials = a.flip_blocks; }
    if (ha.count == kMaxRangeAtts) { launch_histograms(ha, s); ha.count = 0; }
    pin_off[i] = a.slab_off;
  }
  ha.orient = fused_orient;
  launch_histograms(ha, s);
  // scratch words, ranges, histograms and orientation summaries of every attribute: one copy (the pinned buffer mirrors the slab)
  if (!plan_only && !job->dev_tables) HIP_TRY(hipMemcpyAsync(pinned, job->slab.p, job->slab.bytes, hipMemcpyDeviceToHost, s));
  if (timed) HIP_TRY(hipEventRecord(job->ev[3], s));
  return DMI_OK;
}

int encode_phase_b(dmi_job* job, bool plan_only, bool host_chains) {   // host: table normalisation; device: coding records; fills job->run.descs
  // host_chains: the streams are coded on host cores (encode_tail_host): no coding records are built, orientation flags are compacted to bits
  // plan_only: a step sink is set — launches are collected, uploads are deferred to job->run.pending: no HIP call is made
  hipStream_t s = job->stream;
  job->run.pending.clear();
  auto upload_table = [&](void* dst, const void* src, size_t bytes) -> int {
    if (plan_only) { job->run.pending.push_back({dst, src, (bytes + 15) & ~(size_t)15}); return DMI_OK; }
    HIP_TRY(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, s));
    return DMI_OK;
  };
  const uint32_t n_atts = (uint32_t)job->atts.size();
  uint8_t* pinned = job->readback ? job->readback : static_cast<uint8_t*>(job->pinned);
  // ---- stage 4 (host): normalise tables, build chain descriptors -----------------------------------------
  std::vector<ChainDesc>& descs = job->run.descs;
  descs.clear();
  std::vector<AuxInfo>& aux = job->run.aux;
  aux.assign(n_atts, AuxInfo{});
  job->run.hdr_ptr.assign(n_atts, nullptr);
  job->run.hdr_len.assign(n_atts, 0);
  const std::vector<size_t>& pin_off = job->run.pin_off;
  std::string err;
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    const uint8_t* base = pinned + pin_off[i];
    const uint32_t* small = reinterpret_cast<const uint32_t*>(base);
    if (dbg_on(DMI_DBG_TRACE)) std::fprintf(stderr, "[dmi] attribute %u small: %08x %08x %u %u %u %u %u %u\n", i, small[0], small[1], small[2], small[3], small[4], small[5], small[6], small[7]);
    if (small[4]) return fail(DMI_ERR_ZERO_NORMAL, "attribute " + std::to_string(i) + " contains a zero-length normal (reference assert, geom.rs:45)");
    if (small[5]) return fail(DMI_ERR_ALPHABET_TOO_LARGE, "symbol outside the histogram bound");
    { const int brc = check_value_bounds(a, small, i); if (brc) return brc; }
    const uint32_t n = job->tables[a.table].n_seq;
    if (a.n_sym == 0) return fail(DMI_ERR_ENTROPY, "attribute " + std::to_string(i) + " has no values to code (empty histogram)");
    const uint32_t* hist = reinterpret_cast<const uint32_t*>(base + 128);
    int rc = a.ft.build(hist, a.bins, err);
    if (rc) return fail(rc, err);
    job->run.hdr_ptr[i] = a.ft.header.data();
    job->run.hdr_len[i] = (uint32_t)a.ft.header.size();
    std::vector<RansEntry>& rt = a.rt_host;
    rt.assign((a.ft.freq.size() + 3) & ~(size_t)3, RansEntry{0u, 0u, 0u, 0u, 0u});   // 4 entries = 80 bytes = whole 16-byte words
    for (size_t k = 0; k < a.ft.freq.size(); ++k) rt[k] = make_rans_entry(a.ft.freq[k], a.ft.cum[k], a.ft.precision);
    if (!host_chains) {
      { const int urc = upload_table(a.rtable.p, rt.data(), rt.size() * sizeof(RansEntry)); if (urc) return urc; }
      // symbols → coding records in coding order (data-parallel), consumed by the scalar chain
      launch_rans_prep(a.sym.p, a.sym16, a.n_sym, a.rtable.as<RansEntry>(), a.bins, a.rec.as<RansEntry>(), a.batch_flags.as<uint32_t>(), s);
    }
    ChainDesc d{};
    d.kind = 0; d.precision = a.ft.precision; d.n = a.n_sym; d.sym = a.sym.p; d.table = a.rec.as<RansEntry>(); d.state0 = 4u << a.ft.precision; d.batch_flags = a.batch_flags.as<uint32_t>();
    {   // which step the stream's walker uses: the one-byte step pays off when few batches of 64 hold a rare symbol (f < 2^(P-8))
      uint64_t rare = 0;
      for (size_t k = 0; k < a.ft.freq.size(); ++k)
        if (a.ft.freq[k] && ((uint64_t)a.ft.freq[k] << 8) < ((uint64_t)1 << a.ft.precision)) rare += hist[k];
      const double clean = std::pow(1.0 - (double)rare / (double)std::max<uint64_t>(a.n_sym, 1), 64.0);
      d.one_byte = clean > 0.8 ? 1u : 0u;
    }
    d.out = a.out.as<uint8_t>(); d.cap = a.out_cap; d.out_len = a.small.as<uint32_t>() + 8; d.ticks = a.small.as<uint32_t>() + 12;
    aux[i].rans_desc = (int)descs.size();
    descs.push_back(d);
    if (a.scheme == kNormal) {
      // mesh_normal_prediction.rs:147-150
      const uint32_t count_false = small[2];
      aux[i].zero_prob = zero_probability(count_false, (float)n);
      aux[i].count = n;
      ChainDesc r{};
      if (!host_chains) {   // rABS (rans.rs:91-108): bit 1 codes with f1 = 256 - p0 and offset 0, bit 0 with p0 and offset f1
        const uint32_t p0 = aux[i].zero_prob, f1 = 256u - p0;
        launch_bits_prep(a.aux.as<uint8_t>(), n, make_rans_entry(p0, f1, 8), make_rans_entry(f1, 0, 8), a.aux_rec.as<RansEntry>(), s);
        launch_batch_flags(a.aux_rec.as<RansEntry>(), n, nullptr, a.aux_flags.as<uint32_t>(), s);
      }
      r.kind = 1; r.n = n; r.precision = 8; r.state0 = 4096; r.table = a.aux_rec.as<RansEntry>(); r.force_generic = 0; r.batch_flags = a.aux_flags.as<uint32_t>(); r.out = a.aux_out.as<uint8_t>(); r.cap = a.aux_cap; r.out_len = a.small.as<uint32_t>() + 10; r.ticks = a.small.as<uint32_t>() + 13;
      aux[i].desc = (int)descs.size();
      descs.push_back(r);
    } else if (a.scheme == kTexCoord) {
      // stitch per-block summaries: len = Σ count, freq_count_0 = forward transitions with last := true
      const uint32_t nb = orient_summary_blocks(n);
      const uint32_t* sm = reinterpret_cast<const uint32_t*>(base + 128 + (size_t)a.bins_cap * 4);
      uint64_t len = 0, trans = 0;
      uint32_t last = 1;
      for (uint32_t b = 0; b < nb; ++b) {
        const uint32_t cnt = sm[4 * b], first = sm[4 * b + 1], lastv = sm[4 * b + 2], tr = sm[4 * b + 3];
        if (!cnt) continue;
        if (first != last) ++trans;
        trans += tr;
        last = lastv;
        len += cnt;
      }
      aux[i].zero_prob = zero_probability(trans, (float)len + 0.001f);
      aux[i].count = (uint32_t)len;
      ChainDesc r{};
      {   // compact offsets + successor values per 4096-flag chunk, then flags → coding records on the device
        std::vector<uint32_t>& info = a.info_host;
        info.assign(2 * (size_t)std::max(1u, nb), 0u);
        uint32_t off = 0;
        for (uint32_t b = 0; b < nb; ++b) { info[2 * b] = off; off += sm[4 * b]; }
        uint32_t nextv = 1;   // `true` after the last valid entry
        for (uint32_t b = nb; b-- > 0;) { info[2 * b + 1] = nextv; if (sm[4 * b]) nextv = sm[4 * b + 1]; }
        info.resize((info.size() + 3) & ~(size_t)3, 0u);   // whole 16-byte words (the batch driver copies in uint4)
        { const int urc = upload_table(a.chunk_info.p, info.data(), info.size() * 4); if (urc) return urc; }
        const uint32_t p0 = aux[i].zero_prob, f1 = 256u - p0;
        if (host_chains) launch_orient_bits(a.aux.as<uint8_t>(), n, a.chunk_info.as<uint32_t>(), a.aux_bits.as<uint8_t>(), s);
        else {
          launch_orient_prep(a.aux.as<uint8_t>(), n, a.chunk_info.as<uint32_t>(), make_rans_entry(p0, f1, 8), make_rans_entry(f1, 0, 8), a.aux_rec.as<RansEntry>(), s);
          launch_batch_flags(a.aux_rec.as<RansEntry>(), len, nullptr, a.aux_flags.as<uint32_t>(), s);
        }
      }
      r.kind = 2; r.n = len; r.precision = 8; r.state0 = 4096; r.table = a.aux_rec.as<RansEntry>(); r.force_generic = 0; r.batch_flags = a.aux_flags.as<uint32_t>(); r.out = a.aux_out.as<uint8_t>(); r.cap = a.aux_cap; r.out_len = a.small.as<uint32_t>() + 10; r.ticks = a.small.as<uint32_t>() + 13;
      aux[i].desc = (int)descs.size();
      descs.push_back(r);
    }
  }
  return DMI_OK;
}

// Device form of phase B: one k_tables workgroup per attribute (normalisation, serialised table, coding records, metadata
// parameters, chain descriptors written to desc_base[…]), then the record prep — launches only, nothing waits for the host.
// run.descs keeps a host mirror of the static descriptor fields (stream lengths as the host knows them, capacities).
uint32_t count_streams(const dmi_job* job) {
  uint32_t k = 0;
  for (const auto& a : job->atts) k += (a.scheme == kNormal || a.scheme == kTexCoord) ? 2u : 1u;
  return k;
}
int encode_phase_b_dev(dmi_job* job, ChainDesc* desc_base, ChainDesc* hdr_desc_base, bool host_chains) {
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  std::vector<ChainDesc>& descs = job->run.descs;
  descs.clear();
  std::vector<AuxInfo>& aux = job->run.aux;
  aux.assign(n_atts, AuxInfo{});
  job->run.pending.clear();
  // a single job's table stage is ONE launch (a block per attribute); a batch collects per-attribute steps into its multi-item launch
  const bool grouped = !step_sink_active();
  std::unique_ptr<TableGroup> group(grouped ? new TableGroup() : nullptr);
  if (group) group->count = 0;
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    if (a.n_sym == 0) return fail(DMI_ERR_ENTROPY, "attribute " + std::to_string(i) + " has no values to code (empty histogram)");
  }
  for (int pass = 0; pass < 2; ++pass)
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    const uint32_t n = job->tables[a.table].n_seq;
    if (pass == 1) {   // what depends on the table kernel's outputs
      if (i == 0 && group) { launch_tables_group(*group, s); group->count = 0; }
      if (host_chains) {   // no coding records: the streams are coded on host cores from the symbols, the table and the metadata bits
        if (a.scheme == kTexCoord) launch_orient_bits(a.aux.as<uint8_t>(), n, a.chunk_info.as<uint32_t>(), a.aux_bits.as<uint8_t>(), s);
        continue;
      }
      launch_rans_prep(a.sym.p, a.sym16, a.n_sym, a.rtable.as<RansEntry>(), a.bins, a.rec.as<RansEntry>(), a.batch_flags.as<uint32_t>(), s);
      if (a.scheme == kNormal) {
        launch_bits_prep_dev(a.aux.as<uint8_t>(), n, a.aux_entries.as<RansEntry>(), a.aux_rec.as<RansEntry>(), s);
        launch_batch_flags(a.aux_rec.as<RansEntry>(), n, nullptr, a.aux_flags.as<uint32_t>(), s);
      } else if (a.scheme == kTexCoord) {
        launch_orient_prep_dev(a.aux.as<uint8_t>(), n, a.chunk_info.as<uint32_t>(), a.aux_entries.as<RansEntry>(), a.aux_rec.as<RansEntry>(), s);
        launch_batch_flags(a.aux_rec.as<RansEntry>(), n, a.small.as<uint32_t>() + 15, a.aux_flags.as<uint32_t>(), s);
      }
      continue;
    }
    TableAtt ta{};
    ta.hist = a.hist.as<uint32_t>(); ta.freq = a.freq.as<uint32_t>(); ta.rtable = a.rtable.as<RansEntry>(); ta.hdr = a.hdr.as<uint8_t>(); ta.small = a.small.as<uint32_t>();
    ta.n_sym = a.n_sym; ta.bins = a.bins; ta.hdr_cap = a.hdr_cap;
    aux[i].rans_desc = (int)descs.size();
    ta.desc = desc_base + descs.size();
    ta.sym = a.sym.p; ta.rec = a.rec.as<RansEntry>(); ta.batch_flags = a.batch_flags.as<uint32_t>(); ta.out = a.out.as<uint8_t>(); ta.out_cap = a.out_cap;
    ChainDesc d{};
    d.kind = 0; d.n = a.n_sym; d.out = a.out.as<uint8_t>(); d.cap = a.out_cap; d.out_len = a.small.as<uint32_t>() + 8;
    descs.push_back(d);
    if (a.scheme == kNormal || a.scheme == kTexCoord) {
      ta.aux_kind = a.scheme == kNormal ? 1u : 2u;
      ta.n_entries = n;
      ta.summary = a.summary.as<uint32_t>(); ta.chunk_info = a.chunk_info.as<uint32_t>(); ta.aux_entries = a.aux_entries.as<RansEntry>();
      ta.summary_blocks = a.scheme == kTexCoord ? orient_summary_blocks(n) : 0u;
      aux[i].desc = (int)descs.size();
      ta.aux_desc = desc_base + descs.size();
      ta.aux_rec = a.aux_rec.as<RansEntry>(); ta.aux_flags = a.aux_flags.as<uint32_t>(); ta.aux_out = a.aux_out.as<uint8_t>(); ta.aux_cap = a.aux_cap;
      ChainDesc r{};
      r.kind = ta.aux_kind; r.n = n; r.out = a.aux_out.as<uint8_t>(); r.cap = a.aux_cap; r.out_len = a.small.as<uint32_t>() + 10;
      descs.push_back(r);
    }
    ta.hdr_desc = hdr_desc_base ? hdr_desc_base + i : nullptr;
    if (group) {
      group->a[group->count++] = ta;
      if (group->count == kTableGroup) { launch_tables_group(*group, s); group->count = 0; }
    } else {
      launch_tables(ta, s);
    }
  }
  return DMI_OK;
}

// errors the device reports through an attribute's scratch words (device form; the host form meets them in phase B)
// The packed value layouts (QF_P64 / QF_H32) and the 16-bit symbols narrow what they store; what keeps that exact is the quantizer's bound
// 0 ≤ q < 2^bits (NaN → 0, ±inf → a range end).  The joint min/max every encode computes anyway (small[0..1]) is held to that bound here:
// a violated bound is an error, never a silently truncated field.
int check_value_bounds(const AttJob& a, const uint32_t* small, uint32_t i) {
  if (a.port != kCoordwise || (a.qfmt == QF_I32 && !a.sym16)) return DMI_OK;
  const int32_t mn = (int32_t)small[0], mx = (int32_t)small[1];
  if (mn > mx) return DMI_OK;   // (no entries: the seeds)
  if (mn < 0 || (int64_t)mx > ((int64_t)1 << a.bits) - 1)
    return fail(DMI_ERR_ALPHABET_TOO_LARGE, "attribute " + std::to_string(i) + ": a quantized value lies outside [0, 2^bits) — packed layouts cannot hold it");
  return DMI_OK;
}
int check_device_flags(const uint32_t* small, uint32_t i) {
  if (small[4]) return fail(DMI_ERR_ZERO_NORMAL, "attribute " + std::to_string(i) + " contains a zero-length normal (reference assert, geom.rs:45)");
  if (small[5]) return fail(DMI_ERR_ALPHABET_TOO_LARGE, "symbol outside the histogram bound");
  switch (small[7]) {
    case 0: return DMI_OK;
    case 1: return fail(DMI_ERR_ENTROPY, "empty symbol histogram");
    case 2: return fail(DMI_ERR_ENTROPY, "frequency normalisation overflow");
    case 3: return fail(DMI_ERR_ENTROPY, "frequency normalisation underflow");
    case 4: return fail(DMI_ERR_ENTROPY, "normalised frequency of an occurring symbol is zero (the reference encoder does not terminate on this input)");
    default: return fail(DMI_ERR_ENTROPY, "serialised frequency table exceeds its buffer");
  }
}

static int encode_phase_c1(dmi_job* job) {   // after the chains: async read-back of lengths / error flags
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  uint8_t* pinned = job->readback ? job->readback : static_cast<uint8_t*>(job->pinned);
  if (job->dev_tables) {   // nothing has come back yet: scratch words + quantization ranges of every attribute, 128 bytes each
    job->run.pin_off.assign(n_atts, 0);
    for (uint32_t i = 0; i < n_atts; ++i) {
      job->run.pin_off[i] = job->atts[i].slab_off;
      HIP_TRY(hipMemcpyAsync(pinned + job->run.pin_off[i], job->atts[i].small.p, 128, hipMemcpyDeviceToHost, s));
    }
    return DMI_OK;
  }
  for (uint32_t i = 0; i < n_atts; ++i) HIP_TRY(hipMemcpyAsync(pinned + job->run.pin_off[i], job->atts[i].small.p, 64, hipMemcpyDeviceToHost, s));
  return DMI_OK;
}

static int encode_phase_c2(dmi_job* job) {   // lengths known: async copy of the coded bytes
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  uint8_t* pinned = job->readback ? job->readback : static_cast<uint8_t*>(job->pinned);
  const std::vector<size_t>& pin_off = job->run.pin_off;
  const std::vector<AuxInfo>& aux = job->run.aux;
  auto& rans_off = job->run.rans_off;
  auto& aux_off = job->run.aux_off;
  rans_off.assign(n_atts, 0);
  aux_off.assign(n_atts, 0);
  job->run.rans_ptr.assign(n_atts, nullptr);
  job->run.aux_ptr.assign(n_atts, nullptr);
  job->run.rans_len.assign(n_atts, 0);
  job->run.aux_len.assign(n_atts, 0);
  size_t total = 0;
  std::vector<AuxInfo>& aux_w = job->run.aux;
  if (job->dev_tables) {
    job->run.hdr_ptr.assign(n_atts, nullptr);
    job->run.hdr_len.assign(n_atts, 0);
    job->run.hdr_off.assign(n_atts, 0);
    for (uint32_t i = 0; i < n_atts; ++i) {
      const uint32_t* small = reinterpret_cast<const uint32_t*>(pinned + pin_off[i]);
      int frc = check_device_flags(small, i);
      if (!frc) frc = check_value_bounds(job->atts[i], small, i);
      if (frc) return frc;
      aux_w[i].zero_prob = (uint8_t)small[14];
      aux_w[i].count = small[15];
      job->run.hdr_off[i] = total; total += (small[6] + 15u) & ~15u;
    }
  }
  for (uint32_t i = 0; i < n_atts; ++i) {
    const uint32_t* small = reinterpret_cast<const uint32_t*>(pinned + pin_off[i]);
    if (small[9] || small[11]) return fail(DMI_ERR_ENTROPY, small[9] == 1 || small[11] == 1 ? "rANS state too large" : "coder output capacity exceeded");
    if (dbg_on(DMI_DBG_TRACE)) std::fprintf(stderr, "[dmi] attribute %u: rANS chain %.3f ms (%llu symbols), aux chain %.3f ms\n", i, small[12] * 1e-5, (unsigned long long)job->atts[i].n_sym, small[13] * 1e-5);
    rans_off[i] = total; total += (small[8] + 15u) & ~15u;
    if (aux[i].desc >= 0) { aux_off[i] = total; total += (small[10] + 15u) & ~15u; }
  }
  if (total > job->out_pinned_cap) {
    if (job->out_pinned) (void)hipHostFree(job->out_pinned);
    job->out_pinned = nullptr;
    job->out_pinned_cap = total + total / 4 + 4096;
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&job->out_pinned), job->out_pinned_cap, hipHostMallocDefault));
  }
  for (uint32_t i = 0; i < n_atts; ++i) {
    AttJob& a = job->atts[i];
    const uint32_t* small = reinterpret_cast<const uint32_t*>(pinned + pin_off[i]);
    if (small[8]) HIP_TRY(hipMemcpyAsync(job->out_pinned + rans_off[i], a.out.p, small[8], hipMemcpyDeviceToHost, s));
    if (aux[i].desc >= 0 && small[10]) HIP_TRY(hipMemcpyAsync(job->out_pinned + aux_off[i], a.aux_out.p, small[10], hipMemcpyDeviceToHost, s));
    job->run.rans_ptr[i] = job->out_pinned + rans_off[i];
    job->run.aux_ptr[i] = job->out_pinned + aux_off[i];
    job->run.rans_len[i] = small[8];
    job->run.aux_len[i] = aux[i].desc >= 0 ? small[10] : 0u;
    if (job->dev_tables) {
      if (small[6]) HIP_TRY(hipMemcpyAsync(job->out_pinned + job->run.hdr_off[i], a.hdr.p, small[6], hipMemcpyDeviceToHost, s));
      job->run.hdr_ptr[i] = job->out_pinned + job->run.hdr_off[i];
      job->run.hdr_len[i] = small[6];
    }
  }
  return DMI_OK;
}

// Batch form of c1 + c2: lengths, error flags and bytes come from the packed arena (one table + one byte copy per batch).
int encode_phase_c_packed(dmi_job* job, const PackEntry* table, uint32_t first_desc, const uint8_t* arena_host) {
  const uint32_t n_atts = (uint32_t)job->atts.size();
  const std::vector<AuxInfo>& aux = job->run.aux;
  job->run.rans_ptr.assign(n_atts, nullptr);
  job->run.aux_ptr.assign(n_atts, nullptr);
  job->run.rans_len.assign(n_atts, 0);
  job->run.aux_len.assign(n_atts, 0);
  for (uint32_t i = 0; i < n_atts; ++i) {
    const PackEntry& r = table[first_desc + (uint32_t)aux[i].rans_desc];
    if (r.err) return fail(DMI_ERR_ENTROPY, r.err == 1 ? "rANS state too large" : "coder output capacity exceeded");
    job->run.rans_ptr[i] = arena_host + r.offset;
    job->run.rans_len[i] = r.len;
    if (aux[i].desc >= 0) {
      const PackEntry& x = table[first_desc + (uint32_t)aux[i].desc];
      if (x.err) return fail(DMI_ERR_ENTROPY, x.err == 1 ? "rABS state too large" : "coder output capacity exceeded");
      job->run.aux_ptr[i] = arena_host + x.offset;
      job->run.aux_len[i] = x.len;
    }
  }
  return DMI_OK;
}

int encode_phase_c3(dmi_job* job, dmi_buffer* out) {   // host: splice the attribute section
  const uint32_t n_atts = (uint32_t)job->atts.size();
  uint8_t* pinned = job->readback ? job->readback : static_cast<uint8_t*>(job->pinned);
  const std::vector<size_t>& pin_off = job->run.pin_off;
  const std::vector<AuxInfo>& aux = job->run.aux;
  const auto& rans_ptr = job->run.rans_ptr;
  const auto& aux_ptr = job->run.aux_ptr;
  // ---- stage 6 (host): splice the attribute section (encode/attribute/mod.rs:26-57, attribute_encoder.rs:159-160,344-386)
  // written straight into the caller's buffer: its size is bounded by the parts (< 96 bytes of framing per attribute)
  size_t bound = 16;
  for (uint32_t i = 0; i < n_atts; ++i) bound += 96 + (size_t)job->run.hdr_len[i] + job->run.rans_len[i] + job->run.aux_len[i];
  struct RawSink {
    uint8_t* p; size_t n = 0;
    void u8(uint8_t v) { p[n++] = v; }
    void u32(uint32_t v) { std::memcpy(p + n, &v, 4); n += 4; }   // little-endian host
    void f32(float f) { std::memcpy(p + n, &f, 4); n += 4; }
    void leb128(uint64_t v) { do { uint8_t x = v & 0x7F; v >>= 7; u8(v ? (x | 0x80) : x); } while (v); }
    void bytes(const uint8_t* q, size_t k) {
      if (k >= ((size_t)4 << 20)) {   // a large stream (the position stream of a 10M-triangle mesh is 9 MB): the copy — and the first touch of the output pages — on a few threads
        const size_t parts = std::min<size_t>(8, k >> 21);
        std::vector<dmi::Thread> th;
        for (size_t t = 0; t < parts; ++t) th.emplace_back([=] { const size_t lo = k * t / parts, hi = k * (t + 1) / parts; std::memcpy(p + n + lo, q + lo, hi - lo); });
        for (auto& x : th) x.join();
      } else if (k) {
        std::memcpy(p + n, q, k);
      }
      n += k;
    }
    void bytes(const std::vector<uint8_t>& v) { bytes(v.data(), v.size()); }
  } w{static_cast<uint8_t*>(std::malloc(bound + g_out_prefix)), g_out_prefix};   // (g_out_prefix: room for a one-shot call's header + connectivity in front)
  bound += g_out_prefix;
  if (!w.p) return fail(DMI_ERR_OUT_OF_MEMORY, "malloc");
  job->last_fixups = 0;
  for (uint32_t i = 0; i < n_atts; ++i)
    if (job->atts[i].scheme == kTexCoord && job->atts[i].fused_into >= 0) job->last_fixups += reinterpret_cast<const uint32_t*>(pinned + pin_off[i])[3];
  w.u8((uint8_t)n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) { w.u8((uint8_t)((uint8_t)i - 1)); w.u8(job->atts[i].desc.domain); w.u8(0); }   // Q13
  for (uint32_t i = 0; i < n_atts; ++i) {
    const AttJob& a = job->atts[i];
    w.u8(1); w.u8(a.desc.att_type); w.u8(a.desc.component_type); w.u8(a.desc.num_components); w.u8(0); w.u8((uint8_t)a.desc.unique_id); w.u8((uint8_t)a.port);
  }
  for (uint32_t i = 0; i < n_atts; ++i) {
    const AttJob& a = job->atts[i];
    const uint8_t* base = pinned + pin_off[i];
    // (pinned[off..off+64) holds `small` as read back after phase A — or again after the chains: min/max sit at the same offsets)
    const int32_t* mm = reinterpret_cast<const int32_t*>(base);
    const float* meta = reinterpret_cast<const float*>(base + 64);
    w.u8((uint8_t)a.scheme);
    w.u8((uint8_t)a.transform);
    w.u8(1);   // rans_encoding
    w.bytes(job->run.hdr_ptr[i], job->run.hdr_len[i]);
    const uint32_t rans_len = job->run.rans_len[i], aux_len = job->run.aux_len[i];
    w.leb128(rans_len);
    w.bytes(rans_ptr[i], rans_len);
    ByteSink tinfo;
    if (a.transform == kWrapped) { tinfo.u32((uint32_t)mm[0]); tinfo.u32((uint32_t)mm[1]); }
    else if (a.transform == kOctOrth) { tinfo.u32(255); tinfo.u32(127); }
    if (a.scheme == kNormal) {
      w.bytes(tinfo.b);
      w.u8(aux[i].zero_prob);
      w.leb128(aux_len);
      w.bytes(aux_ptr[i], aux_len);
    } else if (a.scheme == kTexCoord) {
      w.u32(aux[i].count);
      w.u8(aux[i].zero_prob);
      w.leb128(aux_len);
      w.bytes(aux_ptr[i], aux_len);
      w.bytes(tinfo.b);
    } else {
      w.bytes(tinfo.b);
    }
    if (a.port == kCoordwise) {   // quantization_coordinate_wise.rs:56-59
      for (int k = 0; k < a.desc.num_components; ++k) w.f32(meta[k]);
      w.f32(meta[a.desc.num_components]);
      w.u8((uint8_t)a.bits);
    } else if (a.port == kOct) {
      w.u8(8);                     // octahedral_quantization.rs:43
    }
  }
  out->data = w.p;
  out->len = w.n;
  out->cap = bound;
  return DMI_OK;
}

// Hybrid tail of a single-job encode (job->host_chains): after the table stage the symbols, the device-built coding tables, the
// serialised tables and the metadata bits come back into pinned staging (largest attribute first, one event per attribute) and every
// stream is coded by host_rans_chain / host_rabs_chain on its own host core as soon as its attribute has arrived; then the splice.
// The strict dependency chain of one stream is the only stage that leaves the device: a 15M-symbol stream takes ≈ 240 ms on a
// scalar-unit walker and ≈ 40 ms on one 5 GHz core.  Batches (dmi_jobs_encode) keep the device chains.
static int encode_tail_host(dmi_job* job, dmi_buffer* out, float* chain_ms, float* longest_ms, float* wait_ms) {
  hipStream_t s = job->stream;
  const uint32_t n_atts = (uint32_t)job->atts.size();
  const bool dev = job->dev_tables;
  struct Slot { size_t small = 0, sym = 0, table = 0, hdr = 0, bits = 0; };
  std::vector<Slot> slot(n_atts);
  size_t need = 0;
  auto take = [&](size_t bytes) { const size_t at = need; need = (need + bytes + 255) & ~(size_t)255; return at; };
  for (uint32_t i = 0; i < n_atts; ++i) {
    const AttJob& a = job->atts[i];
    const bool has_aux = a.scheme == kNormal || a.scheme == kTexCoord;
    slot[i].small = take(128);
    slot[i].sym = take((size_t)a.n_sym * (a.sym16 ? 2 : 4));
    if (dev) { slot[i].table = take((size_t)a.bins * sizeof(RansEntry)); slot[i].hdr = take(a.hdr_cap); }
    if (has_aux) slot[i].bits = take((size_t)job->tables[a.table].n_seq + 16);
  }
  if (!job->stage || job->stage->cap < need) {
    release_stage(job->stage);
    job->stage = acquire_stage(job->cfg.device, need);
    if (!job->stage) return fail(DMI_ERR_OUT_OF_MEMORY, "hipHostMalloc (host-chain staging)");
  }
  uint8_t* base = job->stage->p;
  while (job->copy_ev.size() < n_atts) { hipEvent_t e; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); job->copy_ev.push_back(e); }
  while (job->host_out.size() < 2 * (size_t)n_atts) job->host_out.emplace_back(new HostChainOut());
  std::vector<uint32_t> order(n_atts);
  for (uint32_t i = 0; i < n_atts; ++i) order[i] = i;
  std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return job->atts[x].n_sym > job->atts[y].n_sym; });
  for (uint32_t i : order) {
    AttJob& a = job->atts[i];
    const uint32_t n = job->tables[a.table].n_seq;
    if (dev) {
      HIP_TRY(hipMemcpyAsync(base + slot[i].small, a.small.p, 128, hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(base + slot[i].table, a.rtable.p, (size_t)a.bins * sizeof(RansEntry), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipMemcpyAsync(base + slot[i].hdr, a.hdr.p, a.hdr_cap, hipMemcpyDeviceToHost, s));
    }
    if (a.n_sym) HIP_TRY(hipMemcpyAsync(base + slot[i].sym, a.sym.p, (size_t)a.n_sym * (a.sym16 ? 2 : 4), hipMemcpyDeviceToHost, s));
    if (a.scheme == kNormal && n) HIP_TRY(hipMemcpyAsync(base + slot[i].bits, a.aux.p, n, hipMemcpyDeviceToHost, s));
    if (a.scheme == kTexCoord && n) HIP_TRY(hipMemcpyAsync(base + slot[i].bits, a.aux_bits.p, n, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipEventRecord(job->copy_ev[i], s));
  }
  // streams, longest first; a few host threads pull them
  struct Stream { uint32_t att; bool aux; uint64_t n; };
  std::vector<Stream> streams;
  for (uint32_t i : order) {
    const AttJob& a = job->atts[i];
    streams.push_back({i, false, a.n_sym});
    if (a.scheme == kNormal || a.scheme == kTexCoord) streams.push_back({i, true, job->tables[a.table].n_seq});
  }
  std::stable_sort(streams.begin(), streams.end(), [](const Stream& x, const Stream& y) { return x.n > y.n; });
  std::vector<int> rcs(streams.size(), DMI_OK);
  std::vector<std::string> errs(streams.size());
  std::vector<AuxInfo>& aux = job->run.aux;
  const int device = job->cfg.device;
  std::atomic<size_t> next{0};
  const auto t_chain0 = std::chrono::steady_clock::now();
  auto work = [&] {
    (void)hipSetDevice(device);
    for (size_t k; (k = next.fetch_add(1)) < streams.size();) {
      const Stream& st = streams[k];
      const uint32_t i = st.att;
      AttJob& a = job->atts[i];
      const auto w0 = std::chrono::steady_clock::now();
      if (hipEventSynchronize(job->copy_ev[i]) != hipSuccess) { rcs[k] = DMI_ERR_HIP; errs[k] = "hipEventSynchronize (host-chain staging)"; continue; }
      const auto w1 = std::chrono::steady_clock::now();
      const uint32_t* small = reinterpret_cast<const uint32_t*>(base + slot[i].small);
      if (dev) {
        int frc = check_device_flags(small, i);
        if (!frc) frc = check_value_bounds(a, small, i);
        if (frc) { rcs[k] = frc; errs[k] = g_last_error; continue; }
      }
      HostChainOut& o = *job->host_out[2 * (size_t)i + (st.aux ? 1 : 0)];
      if (!st.aux) {
        const RansEntry* table = dev ? reinterpret_cast<const RansEntry*>(base + slot[i].table) : a.rt_host.data();
        const uint32_t bins = dev ? a.bins : (uint32_t)a.ft.freq.size();
        const uint32_t precision = dev ? small[12] : a.ft.precision;
        if (a.sym16) host_rans_chain16(reinterpret_cast<const uint16_t*>(base + slot[i].sym), a.n_sym, table, bins, precision, o);
        else host_rans_chain(reinterpret_cast<const uint32_t*>(base + slot[i].sym), a.n_sym, table, bins, precision, o);
      } else {
        const uint32_t p0 = dev ? small[14] : aux[i].zero_prob, f1 = 256u - p0;
        const uint64_t count = dev ? small[15] : aux[i].count;
        const RansEntry e[2] = {make_rans_entry(p0, f1, 8), make_rans_entry(f1, 0, 8)};   // rABS (rans.rs:91-108): bit 0 codes with p0 and offset f1, bit 1 with f1 and offset 0
        host_rabs_chain(base + slot[i].bits, count, e, o);
      }
      if (k == 0) {   // the longest stream
        if (wait_ms) *wait_ms = std::chrono::duration<float, std::milli>(w1 - w0).count();
        if (longest_ms) *longest_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - w1).count();
      }
      if (o.err) {
        rcs[k] = o.err == 2 ? DMI_ERR_OUT_OF_MEMORY : DMI_ERR_ENTROPY;
        errs[k] = o.err == 1 ? (st.aux ? "rABS state too large" : "rANS state too large") : (o.err == 2 ? "malloc (host-chain output)" : "symbol outside the coding table");
      }
    }
  };
  {
    const size_t n_threads = std::max<size_t>(1, std::min<size_t>({streams.size(), (size_t)host_threads(), (size_t)16}));
    std::vector<dmi::Thread> th;
    for (size_t t = 1; t < n_threads; ++t) th.emplace_back(with_debug(work));
    work();
    for (auto& x : th) x.join();
  }
  HIP_TRY(hipStreamSynchronize(s));
  if (chain_ms) *chain_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t_chain0).count();
  for (size_t k = 0; k < streams.size(); ++k) if (rcs[k]) return fail(rcs[k], errs[k]);
  // hand the parts to the splice
  job->run.rans_ptr.assign(n_atts, nullptr); job->run.aux_ptr.assign(n_atts, nullptr);
  job->run.rans_len.assign(n_atts, 0); job->run.aux_len.assign(n_atts, 0);
  if (dev) {
    job->readback = base;
    job->run.pin_off.assign(n_atts, 0);
    job->run.hdr_ptr.assign(n_atts, nullptr);
    job->run.hdr_len.assign(n_atts, 0);
  }
  for (uint32_t i = 0; i < n_atts; ++i) {
    const AttJob& a = job->atts[i];
    const HostChainOut& r = *job->host_out[2 * (size_t)i];
    if (r.len > 0xFFFFFFFFull) return fail(DMI_ERR_ENTROPY, "coded stream exceeds 4 GiB");
    job->run.rans_ptr[i] = r.data; job->run.rans_len[i] = (uint32_t)r.len;
    if (a.scheme == kNormal || a.scheme == kTexCoord) {
      const HostChainOut& x = *job->host_out[2 * (size_t)i + 1];
      job->run.aux_ptr[i] = x.data; job->run.aux_len[i] = (uint32_t)x.len;
    }
    if (dev) {
      const uint32_t* small = reinterpret_cast<const uint32_t*>(base + slot[i].small);
      job->run.pin_off[i] = slot[i].small;
      job->run.hdr_ptr[i] = base + slot[i].hdr;
      job->run.hdr_len[i] = small[6];
      aux[i].zero_prob = (uint8_t)small[14];
      aux[i].count = small[15];
    }
  }
  return encode_phase_c3(job, out);
}

// Phase A as one hipGraph replay (single-job re-encodes and the jobs of a batch that keep their own launches): the ≈9 launches and
// the read-back of a job collapse into a single API call.  Jobs with
// event timing or a ToBits attribute (whose alphabet bound needs a mid-phase host wait) stay on the eager path.
int run_phase_a(dmi_job* job) {
  hipStream_t s = job->stream;
  if (!job->pinned) { HIP_TRY(hipSetDevice(job->cfg.device)); HIP_TRY(hipHostMalloc(&job->pinned, job->pinned_bytes, hipHostMallocDefault)); }   // (not inside a stream capture)
  if (job->graph_a) { HIP_TRY(hipSetDevice(job->cfg.device)); HIP_TRY(hipGraphLaunch(job->graph_a, s)); return DMI_OK; }
  bool eligible = !job->have_events && !job->graph_tried && !job->early;   // (the early stage waits for an event of another stream: not captured)
  for (auto& a : job->atts) if (a.port == kToBits) eligible = false;
  if (!eligible) return encode_phase_a(job);
  job->graph_tried = true;
  HIP_TRY(hipSetDevice(job->cfg.device));
  if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) return encode_phase_a(job);
  const int rc = encode_phase_a(job);
  hipGraph_t graph = nullptr;
  const hipError_t e = hipStreamEndCapture(s, &graph);
  if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
  if (e != hipSuccess || !graph) return encode_phase_a(job);
  hipGraphExec_t exec = nullptr;
  const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (ei != hipSuccess || !exec) return encode_phase_a(job);
  job->graph_a = exec;
  HIP_TRY(hipGraphLaunch(exec, s));
  return DMI_OK;
}

int dmi_job_encode(dmi_job* job, dmi_buffer* out) {
  DebugScope debug_scope(job ? &job->debug : nullptr);
  if (!job || !out) return fail(DMI_ERR_INVALID_ARGUMENT, "null");
  hipStream_t s = job->stream;
  const bool timed = job->have_events;
  const auto wall0 = std::chrono::steady_clock::now();
  job->readback = nullptr;
  // (a failure after the first launch waits for the stream before it returns: the caller may destroy the job at once)
  struct Drain { hipStream_t s; bool armed = true; ~Drain() { if (armed) (void)hipStreamSynchronize(s); } } drain{s};
  int rc = encode_phase_a(job);
  if (rc) return rc;
  auto t_tab0 = std::chrono::steady_clock::now(), t_tab1 = t_tab0;
  const bool host_chains = job->host_chains;
  if (job->dev_tables) {
    // tables, metadata parameters and descriptors on the device: the stream runs from the first kernel to the chains without a host wait
    if ((rc = encode_phase_b_dev(job, job->descs.as<ChainDesc>(), nullptr, host_chains))) return rc;
  } else {
    HIP_TRY(hipStreamSynchronize(s));
    t_tab0 = std::chrono::steady_clock::now();
    if ((rc = encode_phase_b(job, false, host_chains))) return rc;
    if (!host_chains) HIP_TRY(hipMemcpyAsync(job->descs.p, job->run.descs.data(), job->run.descs.size() * sizeof(ChainDesc), hipMemcpyHostToDevice, s));
    t_tab1 = std::chrono::steady_clock::now();
  }
  const std::vector<ChainDesc>& descs = job->run.descs;
  if (timed) HIP_TRY(hipEventRecord(job->ev[4], s));
  float host_chain_ms = 0.0f, longest_ms = 0.0f, wait_ms = 0.0f;
  if (host_chains) {
    // hybrid form: symbols + tables back over PCIe, every stream on a host core, splice
    if ((rc = encode_tail_host(job, out, &host_chain_ms, &longest_ms, &wait_ms))) { (void)hipStreamSynchronize(s); return rc; }
  } else {
    uint64_t longest = 0, total = 0;
    for (const ChainDesc& cd : descs) { longest = std::max<uint64_t>(longest, cd.n); total += cd.n; }
    launch_chains(job->descs.as<ChainDesc>(), nullptr, (uint32_t)descs.size(), reinterpret_cast<uint32_t*>(job->descs.as<ChainDesc>() + job->atts.size() * 2),
                  chain_launch_sparse(longest, total, (uint32_t)descs.size()), s);
  }
  if (timed) HIP_TRY(hipEventRecord(job->ev[5], s));
  if (!host_chains) {
    if ((rc = encode_phase_c1(job))) { (void)hipStreamSynchronize(s); return rc; }
    HIP_TRY(hipStreamSynchronize(s));
    if ((rc = encode_phase_c2(job))) { (void)hipStreamSynchronize(s); return rc; }
    HIP_TRY(hipStreamSynchronize(s));
    if ((rc = encode_phase_c3(job, out))) return rc;
  }

  dmi_timings tm{};
  if (timed) {
    HIP_TRY(hipEventSynchronize(job->ev[5]));
    (void)hipEventElapsedTime(&tm.quantize_ms, job->ev[0], job->ev[1]);
    (void)hipEventElapsedTime(&tm.predict_ms, job->ev[1], job->ev[2]);
    (void)hipEventElapsedTime(&tm.histogram_ms, job->ev[2], job->ev[3]);
    (void)hipEventElapsedTime(&tm.rans_ms, job->ev[4], job->ev[5]);
  }
  if (host_chains) tm.rans_ms = host_chain_ms;   // read-back of symbols / tables + the host-core chains (wall clock)
  tm.table_ms = std::chrono::duration<float, std::milli>(t_tab1 - t_tab0).count();
  if (job->early && job->early->t0 && job->early->t1) { float em = 0; if (hipEventElapsedTime(&em, job->early->t0, job->early->t1) == hipSuccess) tm.early_ms = em; else (void)hipGetLastError(); }
  if (timed && job->dev_tables) (void)hipEventElapsedTime(&tm.table_ms, job->ev[3], job->ev[4]);   // k_tables + record prep on the device
  tm.total_ms = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - wall0).count();
  tm.predict_bytes = job->predict_bytes;
  for (auto& a : job->atts) tm.symbols += a.n_sym;
  tm.num_streams = host_chains ? count_streams(job) : (uint32_t)descs.size();
  tm.host_chains = host_chains ? 1u : 0u;
  tm.longest_stream_ms = longest_ms;
  tm.readback_wait_ms = wait_ms;
  tm.texcoord_fixups = job->last_fixups;
  job->last = tm;
  drain.armed = false;   // every path above ended with a stream synchronisation
  return DMI_OK;
}

// Batch form: every job's data-parallel stages are queued back to back, the host waits once, and all rANS/rABS
// streams of all jobs run in ONE k_chains launch (thousands of wavefronts — the regime where the one-wavefront-
// per-stream coder fills the chip).  All jobs must live on the same device; they are serialised on jobs[0]'s
// stream order-wise by using each job's own stream only when they are the same stream (see dmi_encode_attributes_batch).
// Device + pinned staging of one batch read-back, kept between calls (grow-only; a small pool so that concurrent
int dmi_encode_attributes(const dmi_attribute* atts, const dmi_corner_table* tables, uint32_t n_atts, const uint32_t* seeds, uint32_t n_seeds,
                          const dmi_config* cfg, dmi_buffer* out) {
  DebugScope debug_scope(cfg ? cfg->debug : nullptr);
  dmi_job* job = nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  auto ms = [&] { return std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count(); };
  g_one_shot_call = true;   // (create → encode → destroy: no layout work that only pays over many encodes)
  int rc = dmi_job_create(atts, tables, n_atts, seeds, n_seeds, cfg, &job);
  g_one_shot_call = false;
  if (rc) return rc;
  const float t_create = ms();
  rc = dmi_job_encode(job, out);
  g_last_call = job->last;
  g_last_call.job_create_ms = t_create;
  g_last_call.job_create_device_ms = job->create_device_ms;
  dmi_job_destroy(job);
  g_last_call.call_ms = ms();
  return rc;
}

