#!/usr/bin/env python
"""Assemble the committed profiles/ files of a round from a tools/profile.sh output directory (gpurun_out/prof_<tag>/).
   python tools/make_profiles.py r2"""
import collections, csv, glob, json, os, re, shutil, statistics, sys

tag = sys.argv[1] if len(sys.argv) > 1 else 'r3'
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, 'gpurun_out', 'prof_' + tag)
dst = os.path.join(ROOT, 'profiles')


def short(n):
    return re.sub(r'\(anonymous namespace\)::|mpose::|void ', '', n)


shutil.copy(os.path.join(src, 'summary.txt'), os.path.join(dst, tag + '_rocprofv3_summary.txt'))
for sub, name in (('trace', 'kernel_stats'), ('tail', 'tail_kernel_stats')):
    for f in glob.glob(os.path.join(src, sub, '**', '*kernel_stats.csv'), recursive=True):
        shutil.copy(f, os.path.join(dst, '%s_%s.csv' % (tag, name)))

# ---- soft-argmax kernel by problem size (the stats file mixes the three sizes of tools/prof_tail.py) ----
rows = [r for f in glob.glob(os.path.join(src, 'tail', '**', '*kernel_trace.csv'), recursive=True) for r in csv.DictReader(open(f))]
by = collections.defaultdict(list)
for r in rows:
    dur = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if 'softmax_dsnt' in r['Kernel_Name']:
        k_ = short(r['Kernel_Name']).split('(')[0]
        ta = [t.strip() for t in k_[k_.index('<') + 1:k_.rindex('>')].split(',')] if '<' in k_ else []
        rpw = int(ta[4]) if len(ta) > 4 and ta[4].isdigit() else 1          # rows per workgroup (softmax_dsnt_fwd_k<NV, BI, BO, EXP, RPW, NT>)
        by[(k_, int(r['Grid_Size_X']) // 192 // 17 * rpw)].append(dur)
    elif 'bn_add_softmax_k' in r['Kernel_Name']:      # grid (B, 3, 5) x 256 threads: Grid_Size_X = B * 256
        by[(short(r['Kernel_Name']).split('(')[0], int(r['Grid_Size_X']) // 256)].append(dur)
tail = {}
for (k, B), v in sorted(by.items(), key=lambda kv: -kv[0][1]):
    fused = 'bn_add_softmax_k' in k
    targs = [t.strip() for t in k[k.index('<') + 1:k.rindex('>')].split(',')] if '<' in k else []
    bf16 = (len(targs) > 1 and targs[1] == 'true') if fused else ('false, true, true' in k)      # (bn_add_softmax_k<NV, BF_OUT, ALLJ>)
    nbytes = 3 * B * 17 * 1024 * ((10 if bf16 else 12) if fused else (6 if bf16 else 8)) + (3 * B * 17 * 8 if fused else B * 17 * 12)
    med = statistics.median(v)
    allj = fused and len(targs) > 2 and targs[2] == 'true'
    tail['%sB=%d %s heatmaps' % (('fused with the residual sum (all-joints form): ' if allj else 'fused with the residual sum: ') if fused else '', B,
                                 'bf16' if bf16 else 'fp32')] = {
        'kernel': k, 'launches': len(v), 'median_us': med, 'mean_us': sum(v) / len(v), 'algorithmic_bytes': nbytes,
        'GBps': nbytes / med / 1e3, 'frac_of_8TBps': nbytes / med / 1e3 / 8000.0}
json.dump({'_source': 'rocprofv3 --kernel-trace --stats -- python tools/prof_tail.py (the loops of bench.py::tail_microbench); kernel durations by '
                      'problem size from the trace csv', **tail}, open(os.path.join(dst, tag + '_tail_by_size.json'), 'w'), indent=1)

# ---- per-template PMC averages -> traffic / MFMA-busy json that bench.py reads ----
ctr = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ('pmc_sq', 'pmc_lds', 'pmc_fetch', 'pmc_write'):
    for f in glob.glob(os.path.join(src, sub, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            ctr[(short(r['Kernel_Name']).split('(')[0], int(r['Grid_Size']))][r['Counter_Name']].append(float(r['Counter_Value']))
labels = {   # bench.py kernel tag -> (template, threads in the grid, algorithmic bytes) at B = 32, three column groups; the columns run
    # conv_igemm_k / conv_wgrad_k in their three-product fp16 form with row-group staging (template arguments ..., 2, true): fp32 activations in and out
    # round 4: the regular 128-channel blocks' forward launches and second-3x3 data gradient run conv_h.hip's conv_h2r_k<RN, MODE> on
    # producer-split fp16 planes (4 bytes per element, like fp32); MODE 0 = one pass (f_conv2 AND d_conv2: the PMC average mixes the two,
    # so does its algorithmic figure below), MODE 1 = the fused shortcut as a second pass into a second output (f_in_regular)
    'conv:f_conv2/32x32/128->128': ('conv_h2r_k<2, 0, false>', None, 3 * 32768 * (128 * 4 + 128 * 4) + 3 * 9 * 128 * 128 * 4),
    'conv:d_conv2/32x32/128->128': ('conv_h2r_k<2, 0, false>', None, 3 * 32768 * (128 * 4 + 128 * 4 + 128 * 4) + 3 * 9 * 128 * 128 * 4),
    'conv:f_in_regular/32x32/128->128': ('conv_h2r_k<2, 1, false>', None, 3 * 32768 * (128 * 4 + 2 * 128 * 4) + 3 * 10 * 128 * 128 * 4),
    # round 6: the two-input data gradient on conv_h2r_k<2, 2> (both inputs as planes; + the consumer block's c2 and sc read for its BatchNorm-backward sums)
    'conv:d_in_regular/32x32/128->128': ('conv_h2r_k<2, 2, false>', None, 3 * 32768 * (2 * 128 * 4 + 128 * 4 + 2 * 128 * 4) + 3 * 10 * 128 * 128 * 4),
    'conv:f_conv2/16x16/192->192': ('conv_igemm_k<3, 0, 1, true, 2, true>', None, 3 * 8192 * (192 * 4 + 192 * 4) + 3 * 9 * 192 * 192 * 4),
    'conv:d_conv2/16x16/192->192': ('conv_igemm_k<3, 0, 1, false, 2, true>', None, 3 * 8192 * (192 * 4 + 192 * 8) + 3 * 9 * 192 * 192 * 4),
    'conv:f_in_regular/16x16/192->192': ('conv_igemm_k<3, 1, 1, false, 2, true>', None, 3 * 8192 * (192 * 4 + 2 * 192 * 4) + 3 * 10 * 192 * 192 * 4),
    'conv:d_in_regular/16x16/192->192': ('conv_igemm_k<3, 2, 1, false, 2, true>', None, 3 * 8192 * (2 * 192 * 4 + 192 * 4) + 3 * 10 * 192 * 192 * 4),
    # round 3: the row-of-taps weight gradient (wgrad.hip), template <WK, WN, KB, NB, PRO, PAIR, X1>; algorithmic bytes = the input and the
    # gradient(s) read once + the split-K partial sums written once (n_split x taps x Cin x Cout x 4: 28 / 21 / 9 / 7 splits at B = 32)
    # round 6: both weight gradients of an H2 block read planes (template ..., PL = true): the PMC average mixes the two launches
    'wgrad:f_conv2/32x32/128->128': ('conv_wgrad_rows_k<2, 2, 2, 2, false, true, false, true>', None, 3 * 32768 * 128 * 8 + 3 * 28 * 9 * 128 * 128 * 4),
    'wgrad:f_in_regular/32x32/128->128': ('conv_wgrad_rows_k<2, 2, 2, 2, false, true, false, true>', None, 3 * 32768 * 128 * 12 + 3 * 21 * 10 * 128 * 128 * 4),
    'wgrad:f_conv2/16x16/192->192': ('conv_wgrad_rows_k<2, 2, 3, 1, true, true, false, false>', None, 3 * 8192 * 192 * 8 + 3 * 9 * 9 * 192 * 192 * 4),
    'wgrad:f_in_regular/16x16/192->192': ('conv_wgrad_rows_k<2, 2, 3, 1, false, true, false, false>', None, 3 * 8192 * 192 * 12 + 3 * 7 * 10 * 192 * 192 * 4),
}
out = {'_source': 'rocprofv3 --kernel-trace --pmc <one counter group per run> of `python bench.py --steps 1 --warmup 1 --no-cpu-baseline '
                  '--no-kernel-timing --no-overlap-wgrad --eager --no-inference` (tools/profile.sh %s): per-dispatch averages of the kernel '
                  'template (and grid) the label launches; hbm_bytes = 2 x FETCH_SIZE (16-byte/lane streams are tallied at half their bytes on '
                  'gfx950, MI355X_MICROARCH.md; 1 x for conv.hip\'s conv_wgrad_k, whose gathers are dword loads) + WRITE_SIZE; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs)' % tag}
for label, (templ, grid, alg) in labels.items():
    keys = [k for k in ctr if k[0] == templ and (grid is None or k[1] == grid)]
    if not keys:
        continue
    def avg(c):
        vals = [v for k in keys for v in ctr[k].get(c, [])]
        return sum(vals) / len(vals) if vals else None
    fe, wr, busy, gui = avg('FETCH_SIZE'), avg('WRITE_SIZE'), avg('SQ_VALU_MFMA_BUSY_CYCLES'), avg('GRBM_GUI_ACTIVE')
    conf, idx = avg('SQ_LDS_BANK_CONFLICT'), avg('SQ_LDS_IDX_ACTIVE')
    e = {'kernel': templ + (' grid %d' % grid if grid else ''), 'algorithmic_bytes': alg}
    if fe is not None and wr is not None:
        fx = 1 if templ.startswith('conv_wgrad_k') else 2    # conv_wgrad_k gathers with dword loads: counted in full; everything else streams 16 bytes per lane
        e.update(fetch_size_kb_reported=fe, write_size_kb_reported=wr, fetch_correction=fx, hbm_bytes_per_launch=int(fx * fe * 1024 + wr * 1024))
    if busy is not None and gui:
        e['mfma_busy_frac'] = busy / (gui / 8 * 1024)
        e['shader_clock_GHz_note'] = 'GRBM_GUI_ACTIVE / 8 = %.0f cycles per launch' % (gui / 8)
    if conf is not None and idx:
        e['lds_bank_conflict_frac'] = conf / idx
    out[label] = e
json.dump(out, open(os.path.join(dst, tag + '_pmc_traffic.json'), 'w'), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk in ('mfma_busy_frac', 'hbm_bytes_per_launch', 'algorithmic_bytes', 'lds_bank_conflict_frac')}
                  for k, v in out.items() if isinstance(v, dict)}, indent=1))
print(json.dumps(tail, indent=1))
