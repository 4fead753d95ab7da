#!/usr/bin/env python3
"""Ablation builds of the igemm K loop: which ingredient of a tile step costs what?  Generates variants of csrc/conv.hip
by text substitution (results are numerically WRONG -- timing only), builds margipose_amd/_abl/lib_<variant>.so, and
`tools/bench_conv.py` picks one with MPOSE_LIB=<path>.  Not part of the product.
  usage: python tools/experiments/ablate_conv.py            (build all, here, no GPU needed)
         MPOSE_LIB=margipose_amd/_abl/lib_no_b.so python tools/bench_conv.py   (on the GPU box)"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from margipose_amd import build as B

BODY = '{ stage_row(bk, j); load_a_row(t3, j); }'
SUBS = {
    'base': [],
    'no_b': [('        load_b(nb, s_, rn);\n        side(rn);', '        side(rn);')],
    'no_frag': [('        read_frags(bk ^ 1, 0, afA);                // tile k+1, k-group 0\n', ''),
                ('        read_frags(bk ^ 1, 1, afB);                // tile k+1, k-group 1\n', '')],
    'no_apath': [(BODY, '{ }')],
    'no_aload': [(BODY, '{ stage_row(bk, j); }')],
    'no_split': [('      split4(v, h, m, l);\n      unsigned char* dA = sA + buf',
                  '      h = make_uint2(__float_as_uint(v.x), __float_as_uint(v.y)); m = make_uint2(__float_as_uint(v.z), __float_as_uint(v.w)); l = h;\n      unsigned char* dA = sA + buf')],
}
RT_FALSE = '(a.flags & 0x40000000)'          # a condition the compiler cannot fold; always false at run time
SUBS['fix_no_loop'] = [('for (int k = it_begin; k + 1 < it_end; ++k) {', 'for (int k = it_begin; k + 1 < it_end && %s; ++k) {' % RT_FALSE)]
SUBS['fix_tap_const'] = [('''  const int lane_tap = (int)(threadIdx.x & 63) < MPOSE_MAX_TAPS
      ? *reinterpret_cast<const int*>(&g.cls[cls].taps[(threadIdx.x & 63) < MPOSE_MAX_TAPS ? (threadIdx.x & 63) : 0]) : 0;''',
                          '''  const int lane_tap = (int)(threadIdx.x & 63) < 9
      ? ((((int)(threadIdx.x & 63) / 3 - 1) & 0xff) | ((((int)(threadIdx.x & 63) % 3 - 1) & 0xff) << 8) | ((int)(threadIdx.x & 63) << 16)) : 0;''')]
SUBS['fix_no_masks'] = [('      if ((unsigned)iy < (unsigned)g.IH && (unsigned)ix < (unsigned)g.IW) row_taps[j] |= 1u << t;', '      row_taps[j] |= 1u << t;')]
SUBS['fix_no_store'] = [('            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rs_o, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0);',
                         '            if (%s) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v[r]), rs_o, (int)(voff[r] + (unsigned)(rn * 128)), 0, 0);' % RT_FALSE)]
SUBS['fix_few_masks'] = [('          for (int t = 0; t < n_taps; ++t) mark_tap(t);\n', '')]      # only the first two taps get masks (others read as padding)
SUBS['fix_double_masks'] = [('          for (int t = 0; t < n_taps; ++t) mark_tap(t);\n', '          for (int t = 0; t < n_taps; ++t) mark_tap(t);\n          for (int t = n_taps - 1; t >= 0; --t) mark_tap(t);\n          for (int t = 0; t < n_taps; t += 1) mark_tap((t * 7) % n_taps);\n')]      # the mask loop three times (idempotent): its cost x2 on top
SUBS['fix_no_exchange'] = [('    if (KS > 1) {\n      constexpr int BLK = 16 * 64;', '    if (KS > 1 && %s) {\n      constexpr int BLK = 16 * 64;' % RT_FALSE)]
SUBS['mfma_only'] = SUBS['no_b'] + SUBS['no_frag'] + SUBS['no_apath']
# ---- weight-gradient kernel (conv_wgrad_k): what do the in-register splits and the operand loads cost?
SUBS['wg_no_split'] = [('''#pragma unroll
    for (int q = 0; q < 4; ++q) split2(v[2 * q], v[2 * q + 1], hh[q], mm[q], ll[q]);''',
                        '''#pragma unroll
    for (int q = 0; q < 4; ++q) { hh[q] = __float_as_uint(v[2 * q]); mm[q] = __float_as_uint(v[2 * q + 1]); ll[q] = hh[q]; }''')]
SUBS['wg_no_loads'] = [('          load_g(ln2, nb);\n', ''),
                        ('        if (kb + 1 < KB) { split_x(kb + 1, mask_q, an); load_x(ln1, kb + 1); }\n        else { split_x(0, mask_q1, an); load_x(ln2, 0); }',
                         '        if (kb + 1 < KB) { split_x(kb + 1, mask_q, an); }\n        else { split_x(0, mask_q1, an); }')]
SUBS['wg_mfma_only'] = SUBS['wg_no_split'] + SUBS['wg_no_loads']
SUBS['wg_no_loop'] = [('    for (int q = 0; q < n_groups16; ++q) {\n      const Lane ln2 = next_lane();', '    for (int q = 0; q < n_groups16 && (a.n_split & 0x40000000); ++q) {\n      const Lane ln2 = next_lane();')]
SUBS['wg_no_sched'] = [('        __builtin_amdgcn_sched_barrier(0);        // keep each region', '        // (ablation: no region fence)        // keep each region')]
for _k in (1, 2, 4):                            # forced split-K factor (timing of the heuristic's alternatives)
    SUBS['ks%d' % _k] = [('  return best;\n}', '  return %d;\n}' % _k)]
SUBS['b_only'] = SUBS['no_frag'] + SUBS['no_apath']
SUBS['frag_only'] = SUBS['no_b'] + SUBS['no_apath']

SRC = os.environ.get('CONV_SRC', os.path.join(B.CSRC, 'conv.hip'))      # e.g. a worktree with the row-group patch applied
PREFIX = os.environ.get('ABL_PREFIX', '')
src = open(SRC).read()
if PREFIX == 'rowg_':                       # the same ablations for the row-group K loop of the experiment patch
    side = 'if (j < 9) { stage_piece(bnxt, j, sc_n, sh_n); load_piece(g2, j); }'
    import re
    frag_lines = [(l + '\n', '') for l in dict.fromkeys(re.findall(r'            (?:frag_addr\(b(?:cur|nxt), g[cn]\.t(?: \+ \d)?, fa\);|read_frags_g\([01], fa, af[AB]\);)[^\n]*', src))]
    SUBS = {
        'base': [],
        'no_b': SUBS['no_b'],
        'no_apath': [(side, '{ }')],
        'no_aload': [(side, 'if (j < 9) { stage_piece(bnxt, j, sc_n, sh_n); }')],
        'no_frag': frag_lines,
        'no_loop': [('for (int G = g_begin; G < g_end; ++G) {', 'for (int G = g_begin; G < g_end && %s; ++G) {' % RT_FALSE)],
    }
    SUBS['mfma_only'] = SUBS['no_b'] + SUBS['no_apath'] + SUBS['no_frag']
out_dir = os.path.join(B.PKG_DIR, '_abl')
os.makedirs(out_dir, exist_ok=True)
others = [os.path.splitext(s)[0] + '.o' for s in B.sources() if not s.endswith('conv.hip')]
procs = []
only = sys.argv[1:]
for name, subs in SUBS.items():
    if only and not any((PREFIX + name).startswith(o) for o in only):
        continue
    s = src
    for a, b in subs:
        assert s.count(a) >= 1, (name, a)
        s = s.replace(a, b)
    name = PREFIX + name
    f = os.path.join(B.CSRC, '_abl_%s.hip' % name)         # next to conv.hip so that "common.h" resolves
    open(f, 'w').write(s)
    obj = os.path.join(out_dir, 'conv_%s.o' % name)
    procs.append((name, f, obj, subprocess.Popen([B._hipcc()] + B.HIPCC_FLAGS + ['-c', f, '-o', obj], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
for name, f, obj, p in procs:
    o, _ = p.communicate()
    os.remove(f)
    if p.returncode:
        print(name, 'FAILED\n', o.decode()[-2000:]); continue
    lib = os.path.join(out_dir, 'lib_%s.so' % name)
    subprocess.run([B._hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib, obj] + others, check=True)
    print('built', lib)
