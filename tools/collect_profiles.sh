#!/bin/bash
# One gpurun call: the rocprofv3 passes tools/make_profiles.py turns into profiles/<round>/ (counters in passes of their own).
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_kt $R/gpurun_out/prof_fetch $R/gpurun_out/prof_write
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_kt -- python3 $R/bench.py --steps 5 --warmup 3 --no-cpu-baseline --no-secondary --single-stream > $R/gpurun_out/prof_kt.log 2>&1 && echo "kernel trace done" &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_fetch.log 2>&1 && echo "fetch done" &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-secondary > $R/gpurun_out/prof_write.log 2>&1 && echo "write done"
