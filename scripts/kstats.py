"""Print per-kernel averages from a rocprofv3 rocpd database (gpurun_out/<dir>/run_results.db)."""
import re, sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute('select * from top_kernels'):
    m = re.search(r'(k_[a-z_0-9]+(<\d>)?)', r[0])
    print(f"{(m.group(1) if m else r[0][:40]):36s} calls {r[1]:4d}  avg {r[3]/1000 if r[3] > 1e6 else r[3]:10.1f} {'ms' if r[3] > 1e6 else 'us'}")
