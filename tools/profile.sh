#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats of bench.py + PMC passes on the dominant kernels.
# Outputs under gpurun_out/prof_$1/ ; copy the summaries you want judged into profiles/.
TAG=${1:-r1}
OUT=gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats -f csv -d $OUT/trace -o bench -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --eager --no-inference > $OUT/bench_trace.log 2>&1
rocprofv3 -L > $OUT/counters_list.txt 2>&1
# separate PMC passes (never combined with other tracing domains)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -f csv -d $OUT/pmc_sq -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --eager --no-inference > $OUT/pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -f csv -d $OUT/pmc_lds -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --eager --no-inference > $OUT/pmc_lds.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -f csv -d $OUT/pmc_fetch -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --eager --no-inference > $OUT/pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -f csv -d $OUT/pmc_write -o pmc -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-kernel-timing --no-overlap-wgrad --eager --no-inference > $OUT/pmc_write.log 2>&1
find $OUT -name "*.csv" | head -30
python tools/summarize_profile.py $OUT > $OUT/summary.txt 2>&1
cat $OUT/summary.txt | head -80
