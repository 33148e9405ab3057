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
SOURCES = ['cpg_common.cpp', 'mask_kernels.hip', 'rank_prune.hip', 'igemm_conv.hip', 'conv3x3.hip', 'pointwise.hip', 'bn_kernels.hip', 'grad_pack.hip', 'conv3x3_bf16.hip', 'conv3x3_wino.hip', 'conv3x3_wino_wgrad.hip', 'conv3x3_stem.hip']
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
    if force or procs or _stale(LIB, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build_lib(force='--force' in sys.argv)
    print(LIB)
