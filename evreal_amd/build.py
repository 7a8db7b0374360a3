"""Builds libevreal_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m evreal_amd.build [--force]

One object per source under evreal_amd/csrc/ (so per-file floating-point flags are possible),
linked into evreal_amd/libevreal_hip.so.  The .so is git-ignored but travels to the GPU box.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(HERE, 'libevreal_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')

COMMON = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-I' + os.path.join(ROOT, 'include'),
          '-I' + CSRC, '-Wall', '-Wno-unused-function', '-fno-fast-math']
# bit-exact files: one IEEE rounding per op, no fused multiply-add contraction
PER_FILE = {
    'voxelize.hip': ['-ffp-contract=off'],
    'prepost.hip': ['-ffp-contract=off'],
    'metrics.hip': ['-ffp-contract=off'],
    'color.hip': ['-ffp-contract=off'],
}


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(('.hip', '.cpp')))


def _newer(src, dst, extra):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + extra)


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    headers.append(os.path.join(ROOT, 'include', 'evreal_hip.h'))
    headers.append(os.path.abspath(__file__))
    objs, rebuilt = [], False
    procs = []
    for f in sources():
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f + '.o')
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [HIPCC, '-c'] + COMMON + PER_FILE.get(f, []) + ['-x', 'hip', src, '-o', obj]
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((f, subprocess.Popen(cmd)))
            rebuilt = True
    for f, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {f}')
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-o', LIB]
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
