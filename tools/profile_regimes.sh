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
mv $O/pmc_b1024 $O/pmc_b1024_fetch
# where a wave's cycles go (MI355X_MICROARCH.md, rocprofv3 PMC slots): parked in s_waitcnt / barrier, issue stalls, issuing
run b1024 "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES" --batch 1024 --top-k 100 --no-extras --steps 5 --warmup 2
mv $O/pmc_b1024 $O/pmc_b1024_sq
mv $O/pmc_b1024_fetch $O/pmc_b1024
run b8 "" --batch 8 --no-extras --steps 10 --warmup 2
run b8 "FETCH_SIZE" --batch 8 --no-extras --steps 5 --warmup 2
# configs[3] at one GPU (100M x 768 bf16, 153.6 GB resident): why is its last level slower per row than on 10M rows?
timeout -s KILL 400 rocprofv3 --kernel-trace --stats -d $O/kt_b256_100m -o b256_100m --output-format csv -- python3 bench.py --rows-per-gpu 100000000 --no-extras --steps 4 --warmup 2 > $O/kt_b256_100m.log 2>&1 || true
timeout -s KILL 400 rocprofv3 --pmc FETCH_SIZE GRBM_GUI_ACTIVE -d $O/pmc_b256_100m -o b256_100m --output-format csv -- python3 bench.py --rows-per-gpu 100000000 --no-extras --steps 3 --warmup 1 > $O/pmc_b256_100m.log 2>&1 || true
timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE -d $O/pmc_b256_clk -o b256_clk --output-format csv -- python3 bench.py --no-extras --steps 5 --warmup 2 > $O/pmc_b256_clk.log 2>&1 || true
# configs[1] (1M x 384 fp32, one query per call): kernel timeline of a single synchronous caller
timeout -s KILL 150 rocprofv3 --kernel-trace -d $O/kt_c2 -o c2 --output-format csv -- python3 tools/latency_c2.py quick > $O/kt_c2.log 2>&1 || true
python3 tools/trace_timeline.py $(find $O/kt_c2 -name "c2_kernel_trace.csv" | head -1) 16 > $O/c2_timeline.txt 2>&1 || true
timeout -k 10 300 python3 bench.py > $O/bench_full.log 2>&1
tail -n 1 $O/bench_full.log | cut -c1-300
