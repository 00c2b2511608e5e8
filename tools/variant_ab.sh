#!/bin/bash
# A/B of the whole train step under different colour-head variants (DVGO_SHADE_VARIANT): tools/variant_ab.sh 3 23 3 23
set -o pipefail
for v in "$@"; do
  DVGO_SHADE_VARIANT=$v timeout -k 10 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null > /tmp/variant_ab.json || exit 1
  python - "$v" <<'PY'
import json, sys
d = json.loads(open('/tmp/variant_ab.json').read().strip().splitlines()[-1])
k = d.get('kernels', {})
print('variant', sys.argv[1], 'ms_per_step', round(d['ms_per_step'], 4),
      {n: round(v['avg_ms'] * 1e3, 1) for n, v in k.items() if 'shade' in n or 'brick_acc' in n})
PY
done
