#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of bench.py + PMC passes on the dominant kernels + the tail loop.
# Outputs under gpurun_out/prof_$1/ ; copy the summaries you want judged into profiles/.
TAG=${1:-r3}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BENCH="python bench.py --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --eager --no-inference"
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o bench -- $BENCH --steps 5 --warmup 2 > $OUT/bench_trace.log 2>&1
rocprofv3 --kernel-trace --stats -f csv -d $OUT/tail -o tail -- python tools/prof_tail.py > $OUT/tail_trace.log 2>&1
# separate PMC passes (never combined with other tracing domains)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -f csv -d $OUT/pmc_sq -o pmc -- $BENCH --steps 1 --warmup 1 > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_lds -o pmc -- $BENCH --steps 1 --warmup 1 > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o pmc -- $BENCH --steps 1 --warmup 1 > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o pmc -- $BENCH --steps 1 --warmup 1 > $OUT/pmc_write.log 2>&1
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
head -60 $OUT/summary.txt
