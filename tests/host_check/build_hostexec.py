"""Build libunflow_hostexec.so: the kernel source files of unopticalflow_amd/csrc compiled FOR THE BUILD HOST (-DUNFLOW_HOST_CHECK: lanes are
fibers, tests/host_check/hip_on_host.h) behind the library's own C ABI -- TEST INFRASTRUCTURE.  tests/hostexec.py points ops.py at it inside
a `with` block, so the product's Python (autograd wrappers, Model_flow) runs on CPU tensors with the REAL kernel sources underneath; the
product itself never loads it (unopticalflow_amd/_lib.py knows nothing about it) and still has no CPU path.

What is in it: every kernel source file of the library -- photo.hip, ssim.hip, warp.hip, corr.hip (the LDS-DMA ring kernels with their
hand-issued LDS reads included), warp_corr.hip, elementwise.hip, elementwise_bf16.hip, prepare.hip, optim.hip -- as they are.  What the
host cannot show: anything about timing (counted waits are no-ops: every load has landed when its call returns), occupancy or the
hardware's own rounding of v_rcp / MFMA summation order.  UNFLOW_HOSTEXEC_DIR: build into (and load from) another directory."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, 'unopticalflow_amd', 'csrc')
OUT = os.environ.get('UNFLOW_HOSTEXEC_DIR') or os.path.join(HERE, '_build')
LIB = os.path.join(OUT, 'libunflow_hostexec.so')
CLANG = '/opt/rocm/lib/llvm/bin/clang++'
FLAGS = (os.environ.get('UNFLOW_HOSTEXEC_FLAGS', '-O2').split()) + ['-std=c++20', '-fPIC', '-ffp-contract=off', '-DUNFLOW_HOST_CHECK', '-Wno-unknown-attributes', '-Wno-unknown-pragmas', '-Wno-pass-failed',
         '-I', HERE, '-I', CSRC]
SOURCES = [os.path.join(CSRC, f) for f in ('photo.hip', 'ssim.hip', 'warp.hip', 'corr.hip', 'warp_corr.hip', 'elementwise.hip', 'elementwise_bf16.hip', 'prepare.hip', 'optim.hip')]


def _deps():
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(('.h', '.hip'))]
    d += [os.path.join(CSRC, 'bodies', f) for f in os.listdir(os.path.join(CSRC, 'bodies'))]
    d += [os.path.join(HERE, 'hip_on_host.h')] + [os.path.abspath(__file__), os.path.join(ROOT, 'include', 'unflow_hip.h')]
    return d


def build(verbose=False):
    """-> path of the library (rebuilt when a source is newer), or None without the ROCm clang++ (vector extensions, __bf16)."""
    if not os.path.exists(CLANG):
        return None
    os.makedirs(OUT, exist_ok=True)
    if os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(p) for p in _deps()):
        return LIB
    objs = []
    procs = []
    for src in SOURCES:
        obj = os.path.join(OUT, os.path.splitext(os.path.basename(src))[0] + '.o')
        objs.append(obj)
        procs.append((src, subprocess.Popen([CLANG, *FLAGS, '-x', 'c++', '-c', src, '-o', obj], stderr=subprocess.PIPE, text=True)))
    for src, p in procs:
        err = p.communicate()[1]
        if p.returncode != 0:
            raise RuntimeError('%s:\n%s' % (src, err[-3000:]))
    subprocess.run([CLANG, '-shared', '-o', LIB] + [f for f in FLAGS if f.startswith('-fsanitize')] + objs, check=True)
    return LIB


if __name__ == '__main__':
    print(build(verbose=True))
