#!/bin/bash
# kernel trace of bench.py with given args, per-kernel stats to stdout:  tools/kt_bench.sh --workload lego --steps 20
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_prof -- python3 "$R/bench.py" "$@" --no-cpu-baseline --no-secondary > /tmp/kt_out.txt 2>&1 || { tail -20 /tmp/kt_out.txt; exit 1; }
f=$(find /tmp/kt_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:36]:
    print(f'{r["Name"][:72]:72s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:8.1f} us  {float(r["TotalDurationNs"]) / tot * 100:5.1f} %')
PY
