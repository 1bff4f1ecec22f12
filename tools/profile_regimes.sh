set -e
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r01k
rm -rf $O && mkdir -p $O
timeout -k 10 60 rocprofv3 -L > $O/counters.txt 2>&1 || true
grep -c "MFMA" $O/counters.txt || true
timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/pmc_b1 -o b1 --output-format csv -- python3 bench.py --batch 1 --no-extras --steps 5 --warmup 2 > $O/pmc_b1.log 2>&1
timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_b1024 -o b1024 --output-format csv -- python3 bench.py --batch 1024 --top-k 100 --no-extras --steps 5 --warmup 2 > $O/pmc_b1024.log 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/kt_b1 -o b1 --output-format csv -- python3 bench.py --batch 1 --no-extras --steps 10 --warmup 2 > $O/kt_b1.log 2>&1
timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/kt_b1024 -o b1024 --output-format csv -- python3 bench.py --batch 1024 --top-k 100 --no-extras --steps 10 --warmup 2 > $O/kt_b1024.log 2>&1
find $O -name "*.csv" | head -30
du -sh $O
