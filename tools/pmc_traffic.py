"""Collect HBM traffic per launch of the cost-volume / warp entry points with rocprofv3 PMC passes and write
profiles/r4_pmc_traffic.json (via gpurun_out/r4/; read by bench.py's `roofline.traffic`, keyed by the sha256 of the kernel sources).

    python tools/pmc_traffic.py [entry:level ...]           (on the GPU box; ~3 min for the default six)

FETCH_SIZE and WRITE_SIZE are collected in SEPARATE passes (they do not fit one), no trace domains mixed in.  On gfx950
FETCH_SIZE reports half the bytes of wide coalesced streams (MI355X_MICROARCH.md, HBM): x2 before adding WRITE_SIZE (KB)."""
import csv
import glob
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import sources_sha16   # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tools'))

ENTRIES = [('unflow_corr_bwd', 'L2'), ('unflow_corr_fwd', 'L2'), ('unflow_warp_bwd_fused', 'L2'), ('unflow_warp_bwd', 'L2'), ('unflow_warp_fwd', 'L2'),
           ('unflow_corr_bwd', 'L3'), ('unflow_warp_bwd', 'L3'), ('unflow_ssim_loss_fwd', 'S0'), ('unflow_ssim_loss_bwd', 'S0')]
REPS = 3
SKIP = ('randn', 'distribution', 'elementwise', 'fill', 'Fill', 'copy', 'sin', 'cos', 'mul', 'add', 'stack', 'repeat', 'cat', 'arange')


def one_pass(entry, lvl, counter):
    out = tempfile.mkdtemp(prefix='pmc_')
    env = dict(os.environ, TMPDIR='/tmp')
    subprocess.run(['rocprofv3', '--pmc', counter, '--output-format', 'csv', '-d', out, '--',
                    sys.executable, os.path.join(ROOT, 'tools', 'pmc_entry.py'), entry, lvl, str(REPS)],
                   cwd='/tmp', env=env, check=True, capture_output=True, timeout=300)
    per_kernel = {}
    for f in glob.glob(os.path.join(out, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            name = r['Kernel_Name']
            if 'anonymous namespace' not in name and 'unflow' not in name:      # torch's set-up kernels
                continue
            if 'at::native' in name:
                continue
            short = name.replace('(anonymous namespace)::', '').replace('void ', '').split('(')[0][:70]
            per_kernel.setdefault(short, []).append(float(r['Counter_Value']))
    return per_kernel


def main():
    from tools.microbench import LEVELS
    res = {'sources_sha16': sources_sha16(), 'reps': REPS,
           'method': 'rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes, KB per dispatch summed over the kernels '
                     'of one entry-point call; hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 wide-load correction)',
           'entries': {}}
    entries = [tuple(a.split(':')) for a in sys.argv[1:]] or ENTRIES
    for entry, lvl in entries:
        if lvl.startswith('S'):
            C, h, w = 3, 256 >> int(lvl[1:]), 832 >> int(lvl[1:])
        else:
            C, h, w = LEVELS[lvl]
        fetch, write = one_pass(entry, lvl, 'FETCH_SIZE'), one_pass(entry, lvl, 'WRITE_SIZE')
        f_kb = sum(sum(v) for v in fetch.values()) / REPS
        w_kb = sum(sum(v) for v in write.values()) / REPS
        key = '%s %s' % (entry, [16, C, h, w])
        res['entries'][key] = {'fetch_size_kb_raw': round(f_kb, 1), 'write_size_kb': round(w_kb, 1),
                               'hbm_bytes_per_launch': int((2 * f_kb + w_kb) * 1024),
                               'kernels': {k: {'fetch_kb': round(sum(v) / REPS, 1), 'write_kb': round(sum(write.get(k, [0])) / REPS, 1)}
                                           for k, v in fetch.items()}}
        print(key, res['entries'][key]['hbm_bytes_per_launch'], flush=True)
        rnd = os.environ.get('UNFLOW_ROUND', 'r5')
        os.makedirs(os.path.join(ROOT, 'gpurun_out', rnd), exist_ok=True)        # (after every entry: a late failure keeps what was measured)
        json.dump(res, open(os.path.join(ROOT, 'gpurun_out', rnd, rnd + '_pmc_traffic.json'), 'w'), indent=1)
    # the hash of every measured kernel's instruction stream: bench.py keeps serving this capture to later builds whose kernels for an
    # entry are byte-identical (bench._kernels_unchanged)
    try:
        import isa_hashes
        where = {}
        for f, ks in isa_hashes.tree_hashes(ROOT).items():
            for k, v in ks.items():
                where.setdefault(k, {'file': f, 'sha16': v['sha16']})
        res['kernel_isa'] = {'hipcc': isa_hashes.hipcc_version(), 'flags': ' '.join(isa_hashes.build_flags()),
                             'kernels': {k: where[k] for e in res['entries'].values() for k in e['kernels'] if k in where}}
        json.dump(res, open(os.path.join(ROOT, 'gpurun_out', rnd, rnd + '_pmc_traffic.json'), 'w'), indent=1)
    except Exception as e:                                          # noqa: BLE001
        print('kernel ISA hashes not recorded: %s' % e, flush=True)


if __name__ == '__main__':
    main()
