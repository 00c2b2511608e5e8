#!/bin/bash
# one rocprofv3 --pmc pass of a python tool, per-kernel counter means to stdout:
#   tools/pmc.sh "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY" <kernel name filter> tools/wgrad_ab.py --variants 35 --rounds 2
R=${GRAFT_REPO_ROOT:-/root/repo}
ctrs=$1; filt=$2; shift 2
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_prof
rocprofv3 --pmc $ctrs --output-format csv -d /tmp/pmc_prof -- python3 "$R/$1" "${@:2}" > /tmp/pmc_out.txt 2>&1 || { tail -20 /tmp/pmc_out.txt; exit 1; }
python3 "$R/tools/pmc_table.py" /tmp/pmc_prof $filt
