"""Build libunflow_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

    python -m unopticalflow_amd.build [--force]

The library has no torch dependency: it exports the plain-C entry points declared in
include/unflow_hip.h.  It cross-compiles on a machine without a GPU.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libunflow_hip.so')
SOURCES = ('corr.hip', 'warp.hip', 'ssim.hip', 'photo.hip', 'elementwise.hip', 'elementwise_bf16.hip', 'prepare.hip', 'png_host.cpp')
# -ffp-contract=off: mask / SSIM arithmetic must follow the reference op by op; the kernels call
# fmaf() explicitly where a fused multiply-add is wanted.
FLAGS = ('-O3', '--offload-arch=gfx950', '-fPIC', '-shared', '-std=c++17', '-ffp-contract=off',
         '-fno-slp-vectorize')   # SLP packs the 2-px FMAs into v_pk_fma_f32 + a v_mov per pair: slower than plain v_fmac


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(PKG), 'include', 'unflow_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    cmd = [hipcc, *FLAGS, '-o', LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
