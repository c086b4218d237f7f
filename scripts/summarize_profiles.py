"""Turn the rocprofv3 outputs of scripts/profile_round.sh (gpurun_out/<tag>/{stats,fetch,write,sq,tcp}) into the committed summaries
under profiles/.   usage: summarize_profiles.py <tag> <dir> <steps_in_pmc_runs> [--traffic-only]
  profiles/<tag>_kernel_stats.csv   per-kernel calls / total / average (rocprofv3 --kernel-trace --stats)
  profiles/<tag>_pmc_traffic.csv    per-kernel FETCH_SIZE / WRITE_SIZE per launch + the quantize+predict pass per step
  profiles/<tag>_pmc_sq.csv         per-kernel wave issue / wait split (SQ counters) and L1 request counters
  profiles/<tag>_bench.json         the bench line of the same session (roofline.traffic = this session's counters)
--traffic-only prints the pass's HBM bytes per step (FETCH_SIZE doubled per the gfx950 correction of MI355X_MICROARCH.md + WRITE_SIZE)."""
import collections, csv, glob, json, os, re, sys

tag, base, pmc_steps = sys.argv[1], sys.argv[2], int(sys.argv[3])
traffic_only = "--traffic-only" in sys.argv


def short(n):
    m = re.search(r"(k_[a-z_0-9]+)", n)
    return m.group(1) if m else re.sub(r"\(.*", "", n)[:48]


def find(sub, suffix):
    hits = glob.glob(f"{base}/{sub}/**/*{suffix}", recursive=True)
    return hits[0] if hits else None


def agg(sub, names):
    acc = {n: collections.defaultdict(list) for n in names}
    path = find(sub, "_counter_collection.csv")
    if not path:
        return acc
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] in acc:
            acc[r["Counter_Name"]][short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc


# everything that quantizes or predicts (round 6: the early stage — value ranges + value-order quantization, issued behind the device stage's last read-back — is
# PART of the pass: SURVEY §8d's bytes cover it, so do the durations and the traffic; round 5 listed it beside the pass)
PASS_PREFIXES = ("k_value_ranges", "k_value_quantize_rec", "k_seq_quantize", "k_i32_minmax_final", "k_predict_fused", "k_predict_packed", "k_orient_summary", "k_pred_parallelogram",
                 "k_pred_texcoord", "k_pred_delta", "k_seq_gather_rec", "k_texcoord_fixup")
EARLY_PREFIXES = ("k_value_ranges", "k_value_quantize_rec")
F = agg("fetch", ["FETCH_SIZE"])["FETCH_SIZE"]
W = agg("write", ["WRITE_SIZE"])["WRITE_SIZE"]
has_early = any(n.startswith("k_seq_gather_rec") for n in set(F) | set(W))
tf = tw = 0.0
ef = ew = 0.0
rows = []
for n in sorted(set(F) | set(W)):
    f, w = F.get(n, [0.0]), W.get(n, [0.0])
    fa, wa = sum(f) / len(f) * 1024 / 1e6, sum(w) / len(w) * 1024 / 1e6      # counter unit: KiB
    rows.append(f"{n}, {max(len(f), len(w))}, {fa:.2f}, {2 * fa:.2f}, {wa:.2f}")
    if n.startswith(PASS_PREFIXES):
        tf += sum(f) * 1024 / pmc_steps
        tw += sum(w) * 1024 / pmc_steps
    if has_early and n.startswith(EARLY_PREFIXES):
        ef += sum(f) * 1024 / pmc_steps
        ew += sum(w) * 1024 / pmc_steps
traffic = int(2 * tf + tw)
if traffic_only:
    print(traffic)
    sys.exit(0)

os.makedirs("profiles", exist_ok=True)
stats = find("stats", "_kernel_stats.csv")
out = [f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-batch --no-scopes   ({tag}, MI355X)",
       "# min_us beside avg_us: a kernel that shares the device with another queue's copy shows it in its average, not in its minimum",
       "kernel, calls, total_ms, avg_us, min_us, pct"]
pass_avg = pass_min = 0.0
if stats:
    for r in csv.DictReader(open(stats)):
        n = short(r["Name"])
        out.append(f"{n}, {r['Calls']}, {float(r['TotalDurationNs']) / 1e6:.3f}, {float(r['AverageNs']) / 1e3:.2f}, {float(r['MinNs']) / 1e3:.2f}, {r['Percentage']}")
        if n.startswith(PASS_PREFIXES):
            pass_avg += float(r["AverageNs"]) / 1e3
            pass_min += float(r["MinNs"]) / 1e3
    out.append(f"# quantize+predict pass (every kernel that quantizes or predicts: {', '.join(PASS_PREFIXES[:2] + PASS_PREFIXES[-2:])} …): sum of averages {pass_avg:.1f} us, of minima {pass_min:.1f} us "
               f"= {659959872 / max(pass_avg, 1e-9) / 1e3 / 8000:.3f} / {659959872 / max(pass_min, 1e-9) / 1e3 / 8000:.3f} of 8 TB/s for the 659 959 872 algorithmic bytes of the 10M-triangle workload "
               "(the bench line's hipEvent spans add the gaps between the launches)")
open(f"profiles/{tag}_kernel_stats.csv", "w").write("\n".join(out) + "\n")

lines = [f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps {pmc_steps - 1} --warmup 1 --no-cpu-baseline --no-batch --no-scopes ({tag})",
         "# per-launch averages, MB; counter unit KiB.  fetch_x2 = FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read; exact for the",
         "# streaming kernels, an upper bound for the gather kernels whose access width is uncalibrated — MI355X_MICROARCH.md §HBM)",
         "kernel, launches, fetch_MB, fetch_x2_MB, write_MB"] + rows
lines.append(f"# quantize+predict pass per step (early stage included): fetch {tf / 1e6:.1f} MB raw / {2 * tf / 1e6:.1f} MB doubled, write {tw / 1e6:.1f} MB, total {traffic / 1e6:.1f} MB "
             f"= {traffic / 659959872:.2f} x the 659 959 872 algorithmic bytes")
if has_early:
    lines.append(f"# of which the early stage (value ranges + value-order quantization into records, behind the device stage's read-backs): "
                 f"fetch {ef / 1e6:.1f} MB raw / {2 * ef / 1e6:.1f} MB doubled, write {ew / 1e6:.1f} MB, total {(2 * ef + ew) / 1e6:.1f} MB")
open(f"profiles/{tag}_pmc_traffic.csv", "w").write("\n".join(lines) + "\n")

sq_names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_ACTIVE_INST_VALU", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAVES"]
tcp_names = ["TCP_TCC_READ_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum", "TCP_PENDING_STALL_CYCLES_sum", "GRBM_GUI_ACTIVE"]
SQ, TCP = agg("sq", sq_names), agg("tcp", tcp_names)
kernels = sorted(set().union(*[set(v) for v in SQ.values()], *[set(v) for v in TCP.values()]))
sq_lines = [f"# rocprofv3 --pmc {' '.join(sq_names)} and --pmc {' '.join(tcp_names)} (two passes), per-launch averages ({tag})",
            "# SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves: valu_share = ACTIVE_INST_VALU / WAVE_CYCLES = share of the waves'",
            "# lifetime spent issuing vector ALU work, wait_share = WAIT_ANY / WAVE_CYCLES = parked on s_waitcnt / barriers (memory), issue_stall_share = WAIT_INST_ANY / WAVE_CYCLES",
            "kernel, launches, waves, valu_insts_per_wave, valu_share, wait_share, issue_stall_share, l1_to_l2_read_req, l1_accesses, l1_pending_stall_cycles"]
for k in kernels:
    def avg(d, name):
        v = d[name].get(k, [])
        return sum(v) / len(v) if v else 0.0
    wc, waves = avg(SQ, "SQ_WAVE_CYCLES"), avg(SQ, "SQ_WAVES")
    if wc <= 0 and avg(TCP, "TCP_TOTAL_CACHE_ACCESSES_sum") <= 0:
        continue
    sq_lines.append(f"{k}, {len(SQ['SQ_WAVE_CYCLES'].get(k, []))}, {waves:.0f}, {avg(SQ, 'SQ_INSTS_VALU') / max(waves, 1):.0f}, {avg(SQ, 'SQ_ACTIVE_INST_VALU') / max(wc, 1):.3f}, "
                    f"{avg(SQ, 'SQ_WAIT_ANY') / max(wc, 1):.3f}, {avg(SQ, 'SQ_WAIT_INST_ANY') / max(wc, 1):.3f}, {avg(TCP, 'TCP_TCC_READ_REQ_sum'):.0f}, "
                    f"{avg(TCP, 'TCP_TOTAL_CACHE_ACCESSES_sum'):.0f}, {avg(TCP, 'TCP_PENDING_STALL_CYCLES_sum'):.0f}")
open(f"profiles/{tag}_pmc_sq.csv", "w").write("\n".join(sq_lines) + "\n")

bench_log = f"{base}/bench.log"
if os.path.exists(bench_log):
    js = [l for l in open(bench_log) if l.startswith("{")]
    if js:
        open(f"profiles/{tag}_bench.json", "w").write(js[-1])
        b = json.loads(js[-1])
        print(b["value"], b["roofline"])
print("\n".join(out[:14]))
print(lines[-1])
print("\n".join(sq_lines[3:]))
