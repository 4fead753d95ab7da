#!/bin/bash
# PMC passes over the weight-gradient micro-benchmark (tools/bench_wgrad.py): MFMA-busy, LDS bank conflicts, instruction mix.
# Outputs gpurun_out/pmc_wgrad/*.csv + a per-kernel summary on stdout.
OUT=gpurun_out/pmc_wgrad
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -f csv -d $OUT/sq -o pmc -- python tools/bench_wgrad.py > $OUT/sq.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_LDS -f csv -d $OUT/lds -o pmc -- python tools/bench_wgrad.py > $OUT/lds.log 2>&1
python - <<'PY'
import csv, glob, collections
for sub in ('sq', 'lds'):
    for f in glob.glob('gpurun_out/pmc_wgrad/%s/**/*counter_collection.csv' % sub, recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name'][:70]
            acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        for k, d in acc.items():
            if 'wgrad' not in k: continue
            print(sub, k)
            for c, v in sorted(d.items()): print('    %-28s %.4g' % (c, v))
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in d: print('    mfma_busy = %.3f' % (d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES'] / 4 if False else d['SQ_VALU_MFMA_BUSY_CYCLES'] / d['SQ_BUSY_CYCLES']))
            if 'SQ_LDS_BANK_CONFLICT' in d: print('    lds_conflict_frac = %.3f' % (d['SQ_LDS_BANK_CONFLICT'] / max(d['SQ_LDS_IDX_ACTIVE'], 1)))
PY
