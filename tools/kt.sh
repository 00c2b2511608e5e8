#!/bin/bash
# kernel trace of one python tool on the GPU box, per-kernel stats to stdout:  tools/kt.sh tools/wgrad_ab.py --variants 35
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_prof -- python3 "$R/$1" "${@:2}" > /tmp/kt_out.txt 2>&1 || { tail -20 /tmp/kt_out.txt; exit 1; }
f=$(find /tmp/kt_prof -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print(f'{r["Name"][:80]:80s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  min {float(r["MinNs"]) / 1e3:9.1f}  max {float(r["MaxNs"]) / 1e3:9.1f}')
PY
