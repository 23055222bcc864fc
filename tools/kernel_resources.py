"""Register / LDS / scratch budget of every kernel the library ships, from hipcc's own metadata (no GPU needed):

    python tools/kernel_resources.py [--md profiles/rN_kernel_resources.md]

Columns: arch VGPRs + AGPRs (gfx950: one unified file of 512 per SIMD lane; waves per SIMD = floor(512 / (VGPR + AGPR granule))),
SGPRs, LDS bytes per workgroup (160 KB per CU), scratch bytes per lane (anything > 0 is a spill or a dynamically indexed private
array), and the occupancy those allow for the kernel's workgroup size (from __launch_bounds__ / the launch site is not known here:
waves per SIMD by registers only)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))
import isa_hashes  # noqa: E402


def resources(src):
    with tempfile.TemporaryDirectory() as d:
        asm = os.path.join(d, 'k.s')
        r = subprocess.run([isa_hashes.HIPCC, *isa_hashes.build_flags(), '--cuda-device-only', '-S', src, '-o', asm], capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        text = open(asm).read()
    out = {}
    for m in re.finditer(r'\.amdhsa_kernel (\S+)\n(.*?)\.end_amdhsa_kernel', text, flags=re.S):
        body = m.group(2)
        g = lambda key: int(re.search(r'\.amdhsa_%s (\d+)' % key, body).group(1))
        total, accum = g('next_free_vgpr'), g('accum_offset')
        out[m.group(1)] = {'vgpr': min(total, accum), 'agpr': max(total - accum, 0), 'sgpr': g('next_free_sgpr'),
                           'lds': g('group_segment_fixed_size'), 'scratch': g('private_segment_fixed_size')}
    return out


def waves_per_simd(vgpr, agpr):
    total = vgpr + agpr if agpr else vgpr
    total = (total + 7) // 8 * 8
    return max(1, min(8, 512 // max(total, 1)))


if __name__ == '__main__':
    csrc = os.path.join(ROOT, 'unopticalflow_amd', 'csrc')
    rows = []
    for f in sorted(os.listdir(csrc)):
        if f.endswith('.hip'):
            res = resources(os.path.join(csrc, f))
            names = isa_hashes.demangle(list(res))
            for k, v in res.items():
                n = isa_hashes.short_name(names[k])
                if n.startswith('unflow_zero') and f != 'corr.hip':
                    continue                                     # (the two fill kernels are static in a header: one copy is enough)
                rows.append((f, n, v))
    lines = ['| file | kernel | VGPR | AGPR | SGPR | LDS bytes / WG | scratch bytes / lane | waves / SIMD (registers) |', '|---|---|---|---|---|---|---|---|']
    for f, n, v in rows:
        lines.append('| %s | `%s` | %d | %d | %d | %d | %d | %d |' % (f, n, v['vgpr'], v['agpr'], v['sgpr'], v['lds'], v['scratch'], waves_per_simd(v['vgpr'], v['agpr'])))
    text = '\n'.join(lines)
    if '--md' in sys.argv:
        p = sys.argv[sys.argv.index('--md') + 1]
        with open(p, 'w') as fh:
            fh.write('# Register, LDS and scratch budget of the shipped kernels (%s; `python tools/kernel_resources.py`)\n\n'
                     'From hipcc\'s kernel descriptors with the build\'s flags -- no GPU involved.  %d kernels; with scratch: %s.\n\n%s\n'
                     % (isa_hashes.hipcc_version(), len(rows), ', '.join('`%s`' % n for _, n, v in rows if v['scratch']) or 'none', text))
    print(text)
