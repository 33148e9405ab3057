"""Builds cpg_amd/lib/libcpg_hip.so (gfx950 only) from cpg_amd/csrc with hipcc.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot.  `python -m cpg_amd.build` or
`__graft_entry__.build()`.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
LIB = os.path.join(LIBDIR, 'libcpg_hip.so')
SOURCES = ['cpg_common.cpp', 'mask_kernels.hip', 'rank_prune.hip', 'igemm_conv.hip', 'conv3x3.hip', 'pointwise.hip', 'bn_kernels.hip', 'grad_pack.hip', 'conv3x3_bf16.hip', 'conv3x3_wino.hip', 'conv3x3_wino_wgrad.hip', 'conv3x3_stem.hip', 'conv_stem_s2.hip', 'fc_small.hip']
HEADERS = ['cpg_common.h', 'igemm_core.h', os.path.join('..', '..', 'include', 'cpg_hip.h')]
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wall', '-Wno-unused-function']


def _hipcc():
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def export_map():
    """Linker version script: the shared library exports exactly the functions include/cpg_hip.h declares (cross-file helpers
    such as cpg_conv3x3_wino_run stay internal).  CPG_EXPORT_ALL=1 builds without it -- the kernel A/B tools under tools/ that
    drive single kernels (wino_bench.py, wino_wgrad_bench.py, diag_wino_*.py) need those helpers."""
    import re
    header = open(os.path.join(CSRC, '..', '..', 'include', 'cpg_hip.h')).read()
    names = sorted(set(re.findall(r'\b(cpg_[a-z0-9_]+)\s*\(', header)))
    path = os.path.join(LIBDIR, 'exports.map')
    text = '{\n  global:\n' + ''.join('    %s;\n' % n for n in names) + '  local: *;\n};\n'
    if not os.path.exists(path) or open(path).read() != text:
        with open(path, 'w') as f:
            f.write(text)
    return path


def _export_mode_changed(restricted):
    """True when the existing library was linked in the other export mode (CPG_EXPORT_ALL toggled)."""
    stamp = os.path.join(LIBDIR, 'export_mode')
    mode = 'restricted' if restricted else 'all'
    old = open(stamp).read() if os.path.exists(stamp) else None
    if old != mode:
        with open(stamp, 'w') as f:
            f.write(mode)
        return True
    return False


def build_lib(force=False, verbose=True):
    os.makedirs(LIBDIR, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    procs = []
    for src in SOURCES:
        s = os.path.join(CSRC, src)
        o = os.path.join(LIBDIR, os.path.splitext(src)[0] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            cmd = [hipcc] + FLAGS + ['-x', 'hip', '-c', s, '-o', o]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if out and verbose:
            sys.stdout.write(out.decode(errors='replace'))
        if p.returncode != 0:
            failed = True
            print('FAILED:', src)
    if failed:
        raise RuntimeError('hipcc failed')
    link_deps = list(objs)
    link_flags = []
    if os.environ.get('CPG_EXPORT_ALL') != '1':
        vs = export_map()
        link_deps.append(vs)
        link_flags = ['-Wl,--version-script=' + vs]
    if force or procs or _stale(LIB, link_deps) or _export_mode_changed(bool(link_flags)):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC'] + link_flags + ['-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
    print(LIB)
