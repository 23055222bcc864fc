"""Per-kernel hashes of the gfx950 instruction stream hipcc emits for one source file (labels and comments normalised): what tells a
refactoring that changes no machine code from one that does.

    python tools/isa_hashes.py unopticalflow_amd/csrc/corr.hip [--filter corr] [--json out.json]
    python tools/isa_hashes.py --tree <checkout of a commit> --json tests/golden/isa_validated_r4.json [--note '...']

Used in round 5 to keep the pruned fp32 cost-volume kernels byte-identical to the ones the round-4 GPU suite validated
(tests/golden/isa_validated_r4.json, tests/test_abi.py::test_fp32_cost_volume_kernels_are_the_validated_ones)."""
import hashlib
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = '/opt/rocm/bin/hipcc'


def build_flags():
    """The flags the shipped library is compiled with (unopticalflow_amd/build.py): the hashes are of THAT code."""
    sys.path.insert(0, ROOT)
    try:
        from unopticalflow_amd.build import FLAGS
    finally:
        sys.path.pop(0)
    return list(FLAGS)


def hipcc_version(hipcc=HIPCC):
    out = subprocess.run([hipcc, '--version'], capture_output=True, text=True).stdout
    m = re.search(r'HIP version: (\S+)', out)
    c = re.search(r'clang version (\S+)', out)
    return '%s / clang %s' % (m.group(1) if m else '?', c.group(1) if c else '?')


def kernel_streams(asm_text):
    """{mangled kernel name: [instruction lines]} of a `hipcc -S --cuda-device-only` listing."""
    out, cur, body = {}, None, []
    for line in asm_text.splitlines():
        m = re.match(r'^(_Z\S+):', line)
        if m:
            cur, body = m.group(1), []
            out[cur] = body
            continue
        if cur is None:
            continue
        text = line.split(';')[0].strip()
        if text.startswith('.Lfunc_end'):
            cur = None
            continue
        if not text or (text.startswith('.') and not text.startswith('.LBB')):
            continue
        body.append(re.sub(r'\.LBB\d+_', '.LBB_', text))
    return out


_STREAMS = {}
_SCRATCH = {}


def kernel_scratch(asm_text):
    """{mangled kernel name: scratch bytes per lane} from the `; Kernel info:` comment blocks of the same listing (`; ScratchSize: N`)."""
    out, cur = {}, None
    for line in asm_text.splitlines():
        m = re.match(r'^(_Z\S+):', line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r'^; ScratchSize: (\d+)', line)
        if m and cur is not None:
            out[cur] = int(m.group(1))
            cur = None
    return out


def isa_streams(src, hipcc=HIPCC, extra=()):
    """{mangled kernel name: [instruction lines]} of one source file compiled with the build's flags (cached per process)."""
    key = (os.path.abspath(src), os.path.getmtime(src), tuple(extra))
    if key not in _STREAMS:
        with tempfile.TemporaryDirectory() as d:
            asm = os.path.join(d, 'k.s')
            cmd = [hipcc, *build_flags(), '--cuda-device-only', '-S', src, '-o', asm, *extra]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(r.stderr[-2000:])
            text = open(asm).read()
            _STREAMS[key] = kernel_streams(text)
            _SCRATCH[key] = kernel_scratch(text)
    return _STREAMS[key]


def isa_scratch(src, hipcc=HIPCC, extra=()):
    """{mangled kernel name: scratch bytes per lane} of one source file, from the same (cached) compile as isa_streams()."""
    isa_streams(src, hipcc, extra)
    return _SCRATCH[(os.path.abspath(src), os.path.getmtime(src), tuple(extra))]


def isa_hashes(src, hipcc=HIPCC, extra=()):
    return {k: {'sha16': hashlib.sha256('\n'.join(v).encode()).hexdigest()[:16], 'instructions': len(v)}
            for k, v in isa_streams(src, hipcc, extra).items()}


def demangle(names):
    r = subprocess.run(['c++filt'], input='\n'.join(names), capture_output=True, text=True)
    return dict(zip(names, r.stdout.splitlines()))


def short_name(demangled):
    """`void (anonymous namespace)::corr_fwd_kernel<4, 16>(float const*, ...)` -> `corr_fwd_kernel<4, 16>`."""
    s = demangled.replace('(anonymous namespace)::', '')
    s = re.sub(r'^void ', '', s)
    depth = 0
    for i, ch in enumerate(s):
        depth += ch == '<'
        depth -= ch == '>'
        if ch == '(' and depth == 0:
            return s[:i]
    return s


def tree_hashes(root):
    """{file: {kernel: {sha16, instructions}}} for every csrc/*.hip of a checkout rooted at `root`."""
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(root, 'unopticalflow_amd', 'csrc')
    files = [f for f in sorted(os.listdir(csrc)) if f.endswith('.hip')]
    with ThreadPoolExecutor(max_workers=min(len(files), os.cpu_count() or 2)) as ex:      # one hipcc per file, side by side (corr.hip alone is ~15 s)
        hashes = list(ex.map(lambda f: isa_hashes(os.path.join(csrc, f)), files))
    doc = {}
    for f, h in zip(files, hashes):
        names = demangle(list(h))
        doc[f] = {short_name(names[k]): v for k, v in h.items()}
    return doc


if __name__ == '__main__':
    if '--tree' in sys.argv:                                       # e.g. `git archive <commit> unopticalflow_amd/csrc include | tar -x -C /tmp/x`
        root = sys.argv[sys.argv.index('--tree') + 1]
        doc = {'hipcc': hipcc_version(), 'flags': ' '.join(build_flags()), 'files': tree_hashes(root)}
        if '--note' in sys.argv:
            doc['note'] = sys.argv[sys.argv.index('--note') + 1]
    else:
        src = sys.argv[1]
        flt = sys.argv[sys.argv.index('--filter') + 1] if '--filter' in sys.argv else ''
        h = {k: v for k, v in isa_hashes(src if os.path.isabs(src) else os.path.join(ROOT, src)).items() if flt in k}
        names = demangle(list(h))
        doc = {'hipcc': hipcc_version(), 'files': {os.path.basename(src): {short_name(names[k]): v for k, v in h.items()}}}
    if '--json' in sys.argv:
        with open(sys.argv[sys.argv.index('--json') + 1], 'w') as f:
            json.dump(doc, f, indent=1, sort_keys=True)
    for f, ks in sorted(doc['files'].items()):
        for k, v in sorted(ks.items()):
            print(v['sha16'], '%5d' % v['instructions'], f, k)
