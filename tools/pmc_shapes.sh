#!/bin/bash
# Batch-1024 top-100 last cascade level in both bf16 matrix shapes (k_scan_qreg2 = 32x32x16, k_scan_q16 = 16x16x32), same
# box: kernel trace, the SQ wait / issue breakdown, matrix-pipe busy cycles and the clock (GRBM_GUI_ACTIVE / 8 / duration),
# FETCH_SIZE.  Each counter set is a run of its own (rocprofv3 does not split counters over passes).
#   tools/pmc_shapes.sh <tag>      -> gpurun_out/<tag>/{kt,sq,mfma,fetch}_{16,32}/
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
TAG=${1:-r03a}
O=gpurun_out/$TAG
rm -rf $O && mkdir -p $O
ARGS="--batch 1024 --top-k 100 --path mfma --no-extras --steps 5 --warmup 2"
for shape in 32 16; do
  export CRAG_MFMA_SHAPE=$shape
  timeout -s KILL 150 rocprofv3 --kernel-trace --stats -d $O/kt_$shape -o p --output-format csv -- python3 bench.py $ARGS > $O/kt_$shape.log 2>&1 || exit 1
  timeout -s KILL 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_BUSY_CYCLES -d $O/sq_$shape -o p --output-format csv -- python3 bench.py $ARGS > $O/sq_$shape.log 2>&1 || exit 1
  timeout -s KILL 150 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE -d $O/mfma_$shape -o p --output-format csv -- python3 bench.py $ARGS > $O/mfma_$shape.log 2>&1 || exit 1
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE -d $O/fetch_$shape -o p --output-format csv -- python3 bench.py $ARGS > $O/fetch_$shape.log 2>&1 || exit 1
done
python3 tools/pmc_shapes_summarise.py $TAG
