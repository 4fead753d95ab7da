"""Per-kernel-family comparison of two rocprofv3 *_kernel_stats.csv files (ms per step).  usage: diff_kernel_stats.py A.csv B.csv steps"""
import csv, re, sys, collections
FAMS = ['conv_wgrad', 'conv_planes', 'conv_igemm', 'bn_bwd_reduce_finish', 'bn_bwd_reduce', 'bn_bwd_apply_planes', 'bn_bwd_apply', 'bn_add_planes',
        'bn_add', 'split_planes', 'bn_finalize', 'bn_bwd_coef', 'unpack', 'pack_w', 'copyBuffer', 'combiner', 'pool', 'sgd', 'multi_tensor',
        'softmax', 'stage_loss', 'im2col', 'axis_perm', 'fill', 'elementwise']
def load(p):
    d = collections.defaultdict(lambda: [0, 0.0]); full = {}
    for r in csv.DictReader(open(p)):
        n = r['Name'].replace('mpose::(anonymous namespace)::', '').replace('void ', '')
        n = re.sub(r'\((?!anonymous).*', '', n)
        full[n] = (int(r['Calls']), float(r['TotalDurationNs']) / 1e6)
        f = next((k for k in FAMS if k in n), n[:40])
        d[f][0] += int(r['Calls']); d[f][1] += float(r['TotalDurationNs']) / 1e6
    return d, full
a, fa = load(sys.argv[1]); b, fb = load(sys.argv[2]); steps = float(sys.argv[3]) if len(sys.argv) > 3 else 7
print('%-28s %8s %9s | %8s %9s | %7s' % ('family', 'A calls', 'ms/step', 'B calls', 'ms/step', 'B-A'))
for k in sorted(set(a) | set(b), key=lambda k: -(a[k][1] + b[k][1])):
    print('%-28s %8d %9.3f | %8d %9.3f | %+7.3f' % (k, a[k][0], a[k][1] / steps, b[k][0], b[k][1] / steps, (b[k][1] - a[k][1]) / steps))
print('total %.3f %.3f' % (sum(v[1] for v in a.values()) / steps, sum(v[1] for v in b.values()) / steps))
if len(sys.argv) > 4:
    for nm, f in (('A', fa), ('B', fb)):
        print('--', nm)
        for n, (c, t) in sorted(f.items(), key=lambda kv: -kv[1][1])[:40]:
            print('  %-60s %6d %9.3f ms/step %8.2f us' % (n[:60], c, t / steps, 1e3 * t / c))
