"""Build libunflow_hip.so (the gfx950 kernels + C ABI) in-tree with hipcc.

    python -m unopticalflow_amd.build [--force] [--tuning]

The library has no torch dependency: it exports the plain-C entry points declared in
include/unflow_hip.h.  It cross-compiles on a machine without a GPU.  Sources are compiled to
objects in parallel (one hipcc per file) and linked; only stale objects are rebuilt.

``--tuning`` builds ``libunflow_hip_tuning.so`` with ``-DUNFLOW_TUNING``: the same kernels plus the
environment-driven variant / phase-ablation switches used by ``tools/microbench.py``.  The shipped
library is built without it and never reads the environment.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
LIB = os.path.join(PKG, 'libunflow_hip.so')
LIB_TUNING = os.path.join(PKG, 'libunflow_hip_tuning.so')
HOST_LIB = os.path.join(PKG, 'libunflow_host.so')
TORCH_LIB = os.path.join(PKG, 'libunflow_torch.so')
SOURCES = ('corr.hip', 'warp.hip', 'warp_corr.hip', 'ssim.hip', 'photo.hip', 'elementwise.hip', 'elementwise_bf16.hip',
           'prepare.hip', 'optim.hip', 'png_host.cpp')
HOST_SOURCES = ('png_host.cpp',)
# -ffp-contract=off: mask / SSIM arithmetic must follow the reference op by op; the kernels call
# fmaf() explicitly where a fused multiply-add is wanted.
FLAGS = ('-O3', '--offload-arch=gfx950', '-fPIC', '-std=c++17', '-ffp-contract=off',
         '-fno-slp-vectorize')   # SLP packs the 2-px FMAs into v_pk_fma_f32 + a v_mov per pair: slower than plain v_fmac


def _headers():
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    bodies = os.path.join(CSRC, 'bodies')                          # kernel bodies that a single-scale and a multi-scale kernel both include
    deps += [os.path.join(bodies, f) for f in sorted(os.listdir(bodies)) if f.endswith('.inc')] if os.path.isdir(bodies) else []
    deps.append(os.path.join(os.path.dirname(PKG), 'include', 'unflow_hip.h'))
    deps.append(os.path.abspath(__file__))
    return deps


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(hipcc, src, obj, extra, verbose):
    cmd = [hipcc, *FLAGS, *extra, '-c', src, '-o', obj]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)


def build(force=False, verbose=True, tuning=False):
    """Returns the path of the (re)built library."""
    lib = LIB_TUNING if tuning else LIB
    extra = ('-DUNFLOW_TUNING',) + tuple(os.environ.get('UNFLOW_TUNING_EXTRA_FLAGS', '').split()) if tuning else ()
    objdir = os.path.join(PKG, '_obj_tuning' if tuning else '_obj')
    tag = os.environ.get('UNFLOW_TUNING_TAG', '') if tuning else ''
    if tag:                                              # several variant builds side by side (tools/gpu_r4.sh ssim_variants)
        lib = lib[:-3] + '_' + tag + '.so'
        objdir += '_' + tag
    os.makedirs(objdir, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    hdrs = _headers()
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs, objs = [], []
    for name in srcs:
        src = os.path.join(CSRC, name)
        obj = os.path.join(objdir, os.path.splitext(name)[0] + '.o')
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append((src, obj))
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 2)) as ex:
            for f in [ex.submit(_compile, hipcc, s, o, extra, verbose) for s, o in jobs]:
                f.result()
    if jobs or _newer(lib, objs):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    if not tuning:
        build_host(force, verbose)
        build_torch(force, verbose)
    return lib


def build_host(force=False, verbose=True):
    """libunflow_host.so: the host-only helpers (PNG unfilter) without any HIP dependency, so that DataLoader
    worker processes can load them after a fork without touching the HIP runtime."""
    srcs = [os.path.join(CSRC, s) for s in HOST_SOURCES]
    if force or _newer(HOST_LIB, srcs + _headers()):
        cmd = [os.environ.get('CXX', 'g++'), '-O3', '-fPIC', '-shared', '-std=c++17', '-o', HOST_LIB] + srcs
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return HOST_LIB


def build_torch(force=False, verbose=True):
    """libunflow_torch.so: TORCH_LIBRARY(unflow_hip) registration of the C ABI as dispatcher operators (csrc/torch_ops.cpp).
    Host-only C++ compiled with g++ against the torch headers and linked to libunflow_hip.so (found through $ORIGIN)."""
    src = os.path.join(CSRC, 'torch_ops.cpp')
    if not (force or _newer(TORCH_LIB, [src, LIB] + _headers())):
        return TORCH_LIB
    import torch
    from torch.utils import cpp_extension as ce
    tlib = os.path.join(os.path.dirname(torch.__file__), 'lib')
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    cmd = [os.environ.get('CXX', 'g++'), '-O2', '-fPIC', '-shared', '-std=c++17', '-D__HIP_PLATFORM_AMD__=1', '-DUSE_ROCM=1',
           '-D_GLIBCXX_USE_CXX11_ABI=%d' % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    cmd += ['-I' + d for d in ce.include_paths()] + ['-I' + os.path.join(rocm, 'include'), src, '-o', TORCH_LIB,
            '-L' + tlib, '-ltorch', '-ltorch_cpu', '-lc10', '-lc10_hip', '-ltorch_hip', '-L' + PKG, '-lunflow_hip',
            '-Wl,-rpath,$ORIGIN', '-Wl,-rpath,' + tlib]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.run(cmd, check=True)
    return TORCH_LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv, tuning='--tuning' in sys.argv)
