#!/bin/bash
# PMC passes over tools/bench_igemm.py: where do conv_igemm_k's wave cycles go?  Output: per-kernel table on stdout.
OUT=gpurun_out/pmc_igemm
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -f csv -d $OUT/a -o pmc -- python tools/bench_igemm.py > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU GRBM_GUI_ACTIVE -f csv -d $OUT/b -o pmc -- python tools/bench_igemm.py > $OUT/b.log 2>&1
python - <<'PY'
import csv, glob, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for sub in ('a', 'b'):
    for f in glob.glob('gpurun_out/pmc_igemm/%s/**/*counter_collection.csv' % sub, recursive=True):
        for r in csv.DictReader(open(f)):
            k = re.sub(r'\(anonymous namespace\)::|mpose::|void ', '', r['Kernel_Name']).split('(')[0]
            if 'igemm' not in k: continue
            acc[k][r['Counter_Name'] + ('' if r['Counter_Name'] != 'SQ_WAVE_CYCLES' else '_' + sub)] += float(r['Counter_Value'])
for k, d in acc.items():
    wa, wb = d['SQ_WAVE_CYCLES_a'], d['SQ_WAVE_CYCLES_b']
    print(k)
    print('   of wave cycles: inst-active %.2f  valu(incl mfma issue) %.2f  lds %.2f  vmem %.2f  salu %.2f  misc %.2f | waiting(inst) %.2f  wait-lds %.2f' % (
        d['SQ_ACTIVE_INST_ANY'] / wa, d['SQ_ACTIVE_INST_VALU'] / wa, d['SQ_ACTIVE_INST_LDS'] / wa, d['SQ_ACTIVE_INST_VMEM'] / wb, d['SQ_ACTIVE_INST_SCA'] / wb,
        d['SQ_ACTIVE_INST_MISC'] / wb, d['SQ_WAIT_INST_ANY'] / wa, d['SQ_WAIT_INST_LDS'] / wb))
    print('   mfma busy (of GRBM/8 x 1024 SIMD cycles) %.3f ; VALU insts %.3g SALU insts %.3g' % (d['SQ_VALU_MFMA_BUSY_CYCLES'] / (d['GRBM_GUI_ACTIVE'] / 8 * 1024), d['SQ_INSTS_VALU'], d['SQ_INSTS_SALU']))
PY
