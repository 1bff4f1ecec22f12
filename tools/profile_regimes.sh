#!/bin/bash
# rocprofv3 evidence for the bench regimes (run on the GPU box from the repo root):
#   kernel trace + stats and a separate PMC pass each for the default bench (batch-256), the batch-1 GEMV
#   and the batch-1024 top-100 MFMA regime.  Summaries are then filtered into profiles/ (see DESIGN.md section 5).
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r02a}
O=gpurun_out/$TAG
rm -rf $O && mkdir -p $O
run() { # name, counters ("" = kernel trace), bench args...
    local name=$1 pmc=$2; shift 2
    if [ -z "$pmc" ]; then
        timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/kt_$name -o $name --output-format csv -- python3 bench.py "$@" > $O/kt_$name.log 2>&1
    else
        timeout -s KILL 150 rocprofv3 --pmc $pmc -d $O/pmc_$name -o $name --output-format csv -- python3 bench.py "$@" > $O/pmc_$name.log 2>&1
    fi
}
run b256 "" --no-extras --steps 10 --warmup 3
run b256 "FETCH_SIZE" --no-extras --steps 5 --warmup 2
run b1 "" --batch 1 --path gemv --no-extras --steps 10 --warmup 2
run b1 "FETCH_SIZE" --batch 1 --path gemv --no-extras --steps 5 --warmup 2
run b1024 "" --batch 1024 --top-k 100 --no-extras --steps 10 --warmup 2
run b1024 "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --batch 1024 --top-k 100 --no-extras --steps 5 --warmup 2
mv $O/pmc_b1024 $O/pmc_b1024_mfma
run b1024 "FETCH_SIZE" --batch 1024 --top-k 100 --no-extras --steps 5 --warmup 2
run b8 "" --batch 8 --no-extras --steps 10 --warmup 2
run b8 "FETCH_SIZE" --batch 8 --no-extras --steps 5 --warmup 2
timeout -k 10 300 python3 bench.py > $O/bench_full.log 2>&1
tail -n 1 $O/bench_full.log | cut -c1-300
