#!/bin/bash
# One profiling session of the bench workload on the GPU box (run through gpurun from the repo root):
#   scripts/profile_round.sh <tag>
# kernel trace + stats, the HBM-traffic PMC passes (FETCH_SIZE and WRITE_SIZE cannot share a pass: MI355X_MICROARCH.md, PMC slots), an SQ
# pass (issue / wait split of the waves), then bench.py with roofline.traffic taken from THIS session's counters.  Raw outputs land in
# gpurun_out/<tag>/ (scratch); scripts/summarize_profiles.py writes the summaries that are committed under profiles/.
set -u
tag=${1:-round}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d "$out/stats" -o st --output-format csv -- python3 "$root/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-batch --no-scopes --no-traffic > "$out/stats.log" 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d "$out/fetch" -o f --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-scopes --no-traffic > "$out/fetch.log" 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d "$out/write" -o w --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-scopes --no-traffic > "$out/write.log" 2>&1
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d "$out/sq" -o s --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-scopes --no-traffic > "$out/sq.log" 2>&1
timeout 600 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d "$out/tcp" -o t --output-format csv -- python3 "$root/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-batch --no-scopes --no-traffic > "$out/tcp.log" 2>&1
cd "$root"
traffic=$(python3 scripts/summarize_profiles.py "$tag" "$out" 3 --traffic-only)
echo "quantize+predict pass HBM traffic per step from this session's PMC passes: $traffic bytes"
DMI_ROOFLINE_TRAFFIC=$traffic timeout 1200 python3 bench.py > "$out/bench.log" 2>&1
tail -c 400 "$out/bench.log"
python3 scripts/summarize_profiles.py "$tag" "$out" 3
