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
    'histeq.hip': ['-ffp-contract=off'],
    # (the SLP vectoriser pairs the output-transform adds of different accumulator registers in the epilogue: 128 values live, scratch)
    'wino.hip': ['-fno-slp-vectorize'],
}


# kernels whose accumulators must stay in registers: hipcc has demoted them to scratch more than once while this code
# grew (a branchy unrolled epilogue, dynamic indexing of a register array) -- silently, at a 10x slowdown.  These files
# are compiled with -Rpass-analysis=kernel-resource-usage and the build fails if any of their kernels uses scratch.
NO_SCRATCH = {'conv.hip', 'conv_misc.hip', 'lpips.hip', 'wino.hip'}
# sources compiled more than once: (object tag, extra flags).  conv.hip carries one split arithmetic per object
# (csrc/conv.h arith_mode): mode 2 (f16 + MX-fp8, plus the exact-fp32 kernels) and mode 3 (three f16 products, fp32-grade)
VARIANTS = {'conv.hip': [('', ['-DEVR_ARITH=2']), ('.h3', ['-DEVR_ARITH=3']), ('.m6', ['-DEVR_ARITH=4'])]}


# ... with one measured exception: the fused-prediction instantiation of the Winograd kernel (wino.hip, PRED = true) runs at the
# 256 + 256 register cap and hipcc parks up to 19 loop-invariant set-up dwords (76 B per lane) in scratch -- stored before an item's main loop, reloaded
# in its epilogue, nothing inside the MFMA steps (the accumulators stay in AGPRs: the epilogue reads them with v_accvgpr_read)
SCRATCH_ALLOW = {'wino_f32_kernelILb0ELb0ELi0ELb1E': 96}


def _check_no_scratch(fname, remarks):
    import re
    bad, cur = [], None
    for line in remarks.splitlines():
        m = re.search(r'Function Name: (\S+)', line)
        if m:
            cur = m.group(1)
        m = re.search(r'ScratchSize \[bytes/lane\]: (\d+)', line)
        if m and int(m.group(1)) > max([v for k, v in SCRATCH_ALLOW.items() if cur and k in cur] or [0]):
            bad.append((cur, int(m.group(1))))
    if bad:
        raise RuntimeError(f'{fname}: kernels spill to scratch: ' + ', '.join(f'{k} ({b} B/lane)' for k, b in bad))


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
    for f, tag, vflags in [(f, t, fl) for f in sources() for t, fl in VARIANTS.get(f, [('', [])])]:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f + tag + '.o')
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [HIPCC, '-c'] + COMMON + PER_FILE.get(f, []) + vflags + os.environ.get('EVR_EXTRA_HIPCC_FLAGS', '').split() + ['-x', 'hip', src, '-o', obj]
            check = f in NO_SCRATCH
            if check:
                cmd.append('-Rpass-analysis=kernel-resource-usage')
            if verbose:
                print(' '.join(cmd), flush=True)
            procs.append((f, subprocess.Popen(cmd, stderr=subprocess.PIPE if check else None, text=check or None), check, obj))
            rebuilt = True
    for f, p, check, obj in procs:
        err = p.communicate()[1] if check else None
        if p.wait() != 0:
            if err:     # the remarks carry source excerpts: show the diagnostics only
                lines = err.splitlines()
                keep = [i for i, l in enumerate(lines) if 'error:' in l or 'warning:' in l]
                sys.stderr.write('\n'.join(l for i in keep for l in lines[i:i + 4]) + '\n')
            raise RuntimeError(f'hipcc failed on {f}')
        if check:
            other = [l for l in err.splitlines() if 'warning:' in l]
            if other and verbose:
                sys.stderr.write('\n'.join(other) + '\n')
            try:
                _check_no_scratch(f, err)
            except RuntimeError:
                os.remove(obj)
                raise
    if rebuilt or not os.path.exists(LIB):
        cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC'] + objs + ['-lz', '-lpthread', '-o', LIB]      # (zlib: the PNG writers of hostcodec.cpp)
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
    print(LIB)
