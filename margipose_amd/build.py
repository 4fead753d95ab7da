"""Build libmargipose_hip.so (gfx950 only) in-tree with hipcc.  hipcc cross-compiles without a GPU."""
import glob
import os
import shutil
import subprocess

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, 'csrc')
LIB_PATH = os.path.join(PKG_DIR, 'libmargipose_hip.so')
LOG_PATH = os.path.join(CSRC, 'build.log')      # hipcc's output of the last build, kernel-resource-usage remarks included
HIPCC_FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++20', '-fPIC', '-Wall', '-Wno-unused-function',
               '-Wno-unused-variable', '-Wno-unused-but-set-variable', '-fno-slp-vectorize']


# per-source extra flags.  tail.hip: no fusion of a multiply and an add that the source keeps apart (the soft-argmax row arithmetic is
# shared by two kernels whose results must agree bit for bit; see the comment above row_softmax there)
EXTRA_FLAGS = {'tail.hip': ['-ffp-contract=on']}


def _hipcc():
    exe = shutil.which('hipcc') or '/opt/rocm/bin/hipcc'
    if not os.path.exists(exe):
        raise RuntimeError('hipcc not found: libmargipose_hip.so cannot be built')
    return exe


def sources():
    return sorted(glob.glob(os.path.join(CSRC, '*.hip')))


def _included_sources(src):
    """csrc/*.hip files a source pulls in with #include "x.hip"."""
    out = []
    with open(src) as f:
        for line in f:
            line = line.strip()
            if line.startswith('#include "') and line.endswith('.hip"'):
                out.append(line[len('#include "'):-1])
    return out


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(PKG_DIR, '..', 'include', '*.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile every csrc/*.hip into one shared library next to the package."""
    if not force and not is_stale():
        return LIB_PATH
    objs = []
    procs = []
    # a source is recompiled when it, or any header, is newer than its object (conv.hip's three units take about two minutes each, side by side)
    hdrs = glob.glob(os.path.join(CSRC, '*.h')) + glob.glob(os.path.join(PKG_DIR, '..', 'include', '*.h'))
    t_hdr = max(os.path.getmtime(h) for h in hdrs)
    for src in sources():
        obj = os.path.splitext(src)[0] + '.o'
        # (conv_rn3.hip / conv_rn4.hip are conv.hip compiled again for a part of its instantiations: they age with it)
        t_src = max([os.path.getmtime(src)] + [os.path.getmtime(os.path.join(CSRC, inc)) for inc in _included_sources(src)])
        if not force and os.path.exists(obj) and os.path.getmtime(obj) > max(t_src, t_hdr):
            objs.append(obj)
            continue
        cmd = [_hipcc()] + HIPCC_FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ['-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(obj)
    log = []
    for src, p in procs:
        out, _ = p.communicate()
        text = out.decode(errors='replace')
        if p.returncode != 0:
            raise RuntimeError('hipcc failed on %s:\n%s' % (src, '\n'.join(l for l in text.splitlines() if 'remark:' not in l)))
        log.append('==== %s\n%s' % (os.path.basename(src), text))
        if verbose:
            print('\n'.join(l for l in text.splitlines() if 'remark:' not in l))
    # the log keeps ONE section per source: sections of the sources that were not recompiled are carried over
    sections = {}
    if len(procs) != len(objs) and os.path.exists(LOG_PATH):
        for sec in open(LOG_PATH).read().split('==== ')[1:]:
            sections[sec.split('\n', 1)[0]] = '==== ' + sec.rstrip('\n')
    for entry in log:
        sections[entry.split('\n', 1)[0][5:]] = entry.rstrip('\n')
    with open(LOG_PATH, 'w') as f:
        f.write('\n'.join(sections[k] for k in sorted(sections)) + '\n')
    cmd = [_hipcc(), '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB_PATH] + objs
    subprocess.run(cmd, check=True)
    return LIB_PATH


if __name__ == '__main__':
    print(build(force=True, verbose=True))
