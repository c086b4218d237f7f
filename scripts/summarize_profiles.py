"""Turn the rocprofv3 outputs merged into gpurun_out/ into the committed summaries under profiles/.
usage: summarize_profiles.py <tag> <stats_dir> <fetch_dir> <write_dir> <bench_log> <steps_in_pmc_run>"""
import collections, csv, glob, json, re, sys

tag, stats_dir, fetch_dir, write_dir, bench_log, pmc_steps = sys.argv[1:7]
pmc_steps = int(pmc_steps)

def short(n):
    m = re.search(r"(k_[a-z_0-9]+(<\d>)?)", n)
    return m.group(1) if m else n[:40]

rows = list(csv.DictReader(open((glob.glob(f"{stats_dir}/*/*_kernel_stats.csv") + glob.glob(f"{stats_dir}/*_kernel_stats.csv"))[0])))
out = [f"# rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline   ({tag}, MI355X)", "kernel, calls, total_ms, avg_us, pct"]
for r in rows:
    out.append(f"{short(r['Name'])}, {r['Calls']}, {float(r['TotalDurationNs'])/1e6:.3f}, {float(r['AverageNs'])/1e3:.2f}, {r['Percentage']}")
open(f"profiles/{tag}_kernel_stats.csv", "w").write("\n".join(out) + "\n")

def agg(d, cname):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open((glob.glob(f"{d}/*/*_counter_collection.csv") + glob.glob(f"{d}/*_counter_collection.csv"))[0])):
        if r["Counter_Name"] == cname:
            acc[short(r["Kernel_Name"])].append(float(r["Counter_Value"]))
    return acc
F, W = agg(fetch_dir, "FETCH_SIZE"), agg(write_dir, "WRITE_SIZE")
PASS = {"k_value_ranges", "k_value_ranges_final", "k_seq_quantize", "k_i32_minmax_final", "k_predict_fused", "k_orient_summary",
        # the per-attribute kernels of meshes with seams (not launched by the seam-free bench workload)
        "k_pred_parallelogram_wrapped<3>", "k_face_normals", "k_pred_normal_octorth", "k_pred_texcoord_wrapped"}
lines = [f"# rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py --steps {pmc_steps - 1} --warmup 1 --no-cpu-baseline ({tag})",
         "# per-launch averages, MB; counter unit KiB.  fetch_x2 = FETCH_SIZE doubled (gfx950 reports half of a wide coalesced read; exact for the",
         "# streaming kernels, an upper bound for the gather kernels whose access width is uncalibrated — MI355X_MICROARCH.md §HBM)",
         "kernel, launches, fetch_MB, fetch_x2_MB, write_MB"]
tf = tw = 0.0
for n in sorted(set(F) | set(W)):
    f, w = F.get(n, [0]), W.get(n, [0])
    fa, wa = sum(f) / len(f) * 1024 / 1e6, sum(w) / len(w) * 1024 / 1e6
    lines.append(f"{n}, {len(f)}, {fa:.2f}, {2*fa:.2f}, {wa:.2f}")
    if n in PASS or n.startswith("k_predict_fused") or n.startswith("k_pred_parallelogram"):
        tf += fa * len(f) / pmc_steps
        tw += wa * len(w) / pmc_steps
lines.append(f"# quantize+predict pass per step: fetch {tf:.1f} MB raw / {2*tf:.1f} MB doubled, write {tw:.1f} MB")
open(f"profiles/{tag}_pmc_traffic.csv", "w").write("\n".join(lines) + "\n")
bench = [l for l in open(bench_log) if l.startswith("{")][-1]
open(f"profiles/{tag}_bench.json", "w").write(bench)
print("\n".join(out[:16])); print(lines[-1]); print(json.loads(bench)["roofline"], json.loads(bench)["value"])
