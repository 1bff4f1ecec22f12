# same-box A/B of two builds of the library (tools/bin/lib_<name>.so), each configuration twice
run() { label=$1; b=$2; k=$3
  cp tools/bin/lib_$label.so code-rag_amd/lib/libcoderag_hip.so
  timeout -k 10 100 python bench.py --batch $b --top-k $k --no-extras --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$label', j['config']['batch'], j['config']['top_k'], j['value'], j['ms_per_step'], j['exact_fallback_queries'])" || exit 1
}
for rep in 1 2; do for cfg in "256 10" "1024 100" "256 100" "1024 10" "300 200"; do for v in ${AB:-base tuned}; do run $v $cfg; done; done; done
