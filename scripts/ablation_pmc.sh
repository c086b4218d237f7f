#!/bin/bash
# SQ / TCP counters of the sweep_ablation.py subsets (which instantiation of the fused sweep pays what): two PMC passes.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/ablate_pmc
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAVES -d "$out/sq" -o s --output-format csv -- python3 "$root/scripts/sweep_ablation.py" > "$out/sq.log" 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE -d "$out/tcp" -o t --output-format csv -- python3 "$root/scripts/sweep_ablation.py" > "$out/tcp.log" 2>&1
