"""MFMA utilisation of the conv stacks from hardware counters (north_star: "MFMA utilisation for the conv stacks against chip
peak"), not from FLOP / time.

Input: the counter_collection.csv of ONE `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE
SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_MOPS_BF16` pass over bench.py (no trace domains in that pass), plus the
timed-region kernel table of a separate --kernel-trace run for the durations.  Per kernel family:
    mfma_busy = sum SQ_VALU_MFMA_BUSY_CYCLES / (sum GRBM_GUI_ACTIVE * 4 SIMDs * 256 CUs)      (the formula of rocprof's MfmaUtil)
    mfma_flops = sum MOPS * 512                                                                  (rocprof's MfmaFlops*)
    achieved TFLOP/s = mfma_flops per step / that family's kernel time per step (trace run)
Only the LAST `--steps` steps' worth of dispatches of each kernel are kept (MIOpen's warm-up benchmarking runs other solvers).

    python tools/summarize_mfma.py <counter_collection.csv> <timed_region_stats.csv> <out.json> --steps 9 --peak 157.3
"""
import collections
import csv
import json
import sys


def family(name):
    n = name.lower()
    if 'igemm_fwd' in n: return 'miopen igemm fwd'
    if 'igemm_bwd' in n: return 'miopen igemm bwd-data'
    if 'igemm_wrw' in n: return 'miopen igemm wrw'
    if 'grouped_conv_bwd_data' in n: return 'ck grouped conv bwd-data'
    if 'grouped_conv' in n: return 'ck grouped conv'
    if 'winograd' in n or 'sp3' in n: return 'miopen winograd'
    return None


def main():
    cc, stats, out = sys.argv[1], sys.argv[2], sys.argv[3]
    steps = int(sys.argv[sys.argv.index('--steps') + 1]) if '--steps' in sys.argv else 9
    peak = float(sys.argv[sys.argv.index('--peak') + 1]) if '--peak' in sys.argv else 157.3
    calls = {r['Name']: int(r['Calls']) for r in csv.DictReader(open(stats))}          # dispatches in the timed steps
    dur = {r['Name']: float(r['TotalDurationNs']) for r in csv.DictReader(open(stats))}
    per = collections.defaultdict(lambda: collections.defaultdict(list))                # kernel -> counter -> per-dispatch values, in order
    for r in csv.DictReader(open(cc)):
        per[r['Kernel_Name']][r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    fam = collections.defaultdict(lambda: collections.defaultdict(float))
    for k, ctrs in per.items():
        f = family(k)
        if f is None or k not in calls:
            continue
        n = calls[k]                                     # keep the last n dispatches = the timed steps
        for c, v in ctrs.items():
            v.sort()
            fam[f][c] += sum(x for _, x in v[-n:])
        fam[f]['ns'] += dur[k]
        fam[f]['dispatches'] += n
    import os
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import host_sources_sha16, sources_sha16          # the capture is valid for exactly these sources (bench.py checks)
    res = {'steps': steps, 'peak_tflops': peak, 'sources_sha16': sources_sha16(), 'host_sources_sha16': host_sources_sha16(), 'families': {}}
    tot = collections.defaultdict(float)
    for f, v in sorted(fam.items(), key=lambda kv: -kv[1]['ns']):
        mops = v.get('SQ_INSTS_VALU_MFMA_MOPS_F32', 0.0) + v.get('SQ_INSTS_VALU_MFMA_MOPS_BF16', 0.0)
        flops = mops * 512.0
        busy = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / max(v.get('GRBM_GUI_ACTIVE', 0.0) * 4 * 256, 1.0)
        res['families'][f] = {'dispatches_per_step': round(v['dispatches'] / steps, 1), 'ms_per_step': round(v['ns'] / steps / 1e6, 3),
                              'mfma_busy_frac': round(busy, 4), 'mfma_tflop_per_step': round(flops / steps / 1e12, 4),
                              'achieved_tflops': round(flops / max(v['ns'], 1.0) / 1e3, 1),
                              'frac_of_peak': round(flops / max(v['ns'], 1.0) / 1e3 / peak, 4)}
        res['families'][f]['raw'] = {c: v.get(c, 0.0) for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CU_CYCLES', 'GRBM_GUI_ACTIVE')}
        for c in ('SQ_VALU_MFMA_BUSY_CYCLES', 'GRBM_GUI_ACTIVE', 'ns'):
            tot[c] += v.get(c, 0.0)
        tot['flops'] += flops
    res['conv_kernels_only'] = {'ms_per_step': round(tot['ns'] / steps / 1e6, 3), 'mfma_busy_frac': round(tot['SQ_VALU_MFMA_BUSY_CYCLES'] / max(tot['GRBM_GUI_ACTIVE'] * 1024, 1.0), 4),
                                'mfma_tflop_per_step': round(tot['flops'] / steps / 1e12, 4), 'achieved_tflops': round(tot['flops'] / max(tot['ns'], 1.0) / 1e3, 1),
                                'frac_of_peak': round(tot['flops'] / max(tot['ns'], 1.0) / 1e3 / peak, 4)}
    res['note'] = ('frac_of_peak = hardware-counted MFMA FLOPs (SQ_INSTS_VALU_MFMA_MOPS_* x 512, what rocprof calls MfmaFlops) / kernel time / dense peak -- '
                   'the utilisation figure to quote.  mfma_busy_frac is rocprof\'s MfmaUtil formula on the raw counters (SQ_VALU_MFMA_BUSY_CYCLES / '
                   '(GRBM_GUI_ACTIVE x 4 x 256)); ROCm 7.2 has no gfx950 section for the derived metrics and the busy-cycle counter comes out ~8x below '
                   'the FLOP-derived utilisation on this chip, so it is kept as a raw reading only.')
    json.dump(res, open(out, 'w'), indent=1)
    print(json.dumps(res['conv_kernels_only']))
    for f, v in res['families'].items():
        print('  %-28s %s' % (f, v))


if __name__ == '__main__':
    main()
