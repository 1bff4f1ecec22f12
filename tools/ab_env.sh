#!/bin/bash
# Same-box A/B of one environment variable: tools/ab_env.sh VAR "v1 v2 ..." <bench args...>   (each value twice, interleaved)
set -e
cd "$GRAFT_REPO_ROOT"
var=$1; vals=$2; shift 2
for rep in 1 2; do for v in $vals; do
  env $var=$v timeout -k 10 200 python bench.py --no-extras "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print('$var=$v', 'value', j['value'], 'ms/step', j['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'mfma', r['mfma_tflops'], 'step_mfma_frac', r['step']['mfma_frac'], 'fallback', j['exact_fallback_queries'])"
done; done
