cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/pb; rocprofv3 --kernel-trace --stats -d /tmp/pb -o x --output-format csv -- python3 $GRAFT_REPO_ROOT/scripts/experiments/plain_vs_seams.py > /dev/null 2>&1
grep -E "k_mb_value_insert|k_mb_point_insert|k_conn_faces" /tmp/pb/x_kernel_stats.csv | cut -c1-140
