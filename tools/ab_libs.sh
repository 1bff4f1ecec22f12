#!/bin/bash
# Same-box A/B of library builds: tools/ab_libs.sh "<lib names under tools/bin, without lib_ and .so>" <bench args...>
# Each build runs the bench command twice, interleaved (boxes differ by up to 5 %: only same-box ratios count).
set -e
cd "$GRAFT_REPO_ROOT"
names=$1; shift
cp code-rag_amd/lib/libcoderag_hip.so /tmp/lib_keep.so
for rep in 1 2; do for v in $names; do
  cp tools/bin/lib_$v.so code-rag_amd/lib/libcoderag_hip.so
  timeout -k 10 200 python bench.py --no-extras "$@" 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = j['roofline']
print('$v', 'value', j['value'], 'ms/step', j['ms_per_step'], 'kernel_ms', r['kernel_ms'], 'hbm', round(r['algorithmic_bytes'] / r['kernel_ms'] / 1e6, 1), 'GB/s mfma', r['mfma_tflops'], 'fallback', j['exact_fallback_queries'])" || { cp /tmp/lib_keep.so code-rag_amd/lib/libcoderag_hip.so; exit 1; }
done; done
cp /tmp/lib_keep.so code-rag_amd/lib/libcoderag_hip.so
