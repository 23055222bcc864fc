"""Time individual HIP operators on the pyramid-level shapes of the 832x256, B=8 step (2B=16 for
corr / feature warp).  GPU only.  python tools/microbench.py [op ...]
Backward figures go through torch.autograd.grad, i.e. they include ~40-60 us of host-side autograd dispatch when
the kernel itself is shorter than that (the small loss kernels); bench.py's kernel_survey has the in-step times."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from unopticalflow_amd import ops, _lib   # noqa: E402

if os.environ.get('UNFLOW_MICROBENCH_TUNING') == '1':
    # the tuning build (python -m unopticalflow_amd.build --tuning): same kernels + the environment-driven variant
    # switches the sweeps below flip; the shipped library has none
    from unopticalflow_amd import build as _build
    _lib.LIB_PATH = _build.LIB_TUNING

LEVELS = {'L2': (32, 64, 208), 'L3': (64, 32, 104), 'L4': (96, 16, 52), 'L5': (128, 8, 26), 'L6': (196, 4, 13)}


def timeit(fn, n=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3   # us


def corr8(B=16):
    """BASELINE config 5: d=8 on every pyramid level."""
    corr(B, 8)


def corr(B=16, d=4):
    D2 = (2 * d + 1) ** 2
    for name, (C, h, w) in LEVELS.items():
        f1 = torch.randn(B, C, h, w, device='cuda')
        f2 = torch.randn(B, C, h, w, device='cuda')
        g = torch.randn(B, D2, h, w, device='cuda')
        cv = torch.empty(B, D2, h, w, device='cuda')
        gf1, gf2 = torch.empty_like(f1), torch.empty_like(f2)
        lib = _lib.load()
        P = ops._ptr
        fb = 4 * B * h * w * (2 * C + D2)
        bb = 4 * B * h * w * (4 * C + D2)
        tf = timeit(lambda: lib.unflow_corr_fwd(P(f1), P(f2), P(cv), B, C, h, w, d, ops._stream()))
        tb = timeit(lambda: lib.unflow_corr_bwd(P(f1), P(f2), P(g), P(gf1), P(gf2), B, C, h, w, d, ops._stream()))
        print('corr d=%d %s [%d,%d,%d,%d] variant=%s  fwd %7.1f us (%6.0f GB/s)   bwd %7.1f us (%6.0f GB/s)' % (
            d, name, B, C, h, w, os.environ.get('UNFLOW_CORR_VARIANT', 'auto'), tf, fb / tf / 1e3, tb, bb / tb / 1e3), flush=True)


def _smooth_flow(B, h, w):
    """What a flow network produces: a translation plus low-frequency variation (neighbouring pixels land on
    neighbouring source pixels).  The backward scatter merges adjacent taps in-wave; i.i.d. noise defeats that."""
    yy, xx = torch.meshgrid(torch.arange(h, device='cuda', dtype=torch.float32),
                            torch.arange(w, device='cuda', dtype=torch.float32), indexing='ij')
    u = 2.3 + 1.5 * torch.sin(xx / 37.0) * torch.cos(yy / 23.0)
    v = -1.1 + 0.8 * torch.cos(xx / 29.0 + yy / 41.0)
    return torch.stack((u, v), 0)[None].repeat(B, 1, 1, 1).contiguous()


def warp(B=16):
    for name, (C, h, w) in list(LEVELS.items())[:4]:
        x = torch.randn(B, C, h, w, device='cuda', requires_grad=True)
        g = torch.randn(B, C, h, w, device='cuda')
        fb, bb = 4 * B * h * w * (2 * C + 2), 4 * B * h * w * (3 * C + 4)
        for kind, fl0 in (('smooth flow', _smooth_flow(B, h, w)), ('noise flow ', torch.randn(B, 2, h, w, device='cuda') * 2)):
            fl = fl0.requires_grad_()
            tf = timeit(lambda: ops.warp_flow(x.detach(), fl.detach()))
            y = ops.warp_flow(x, fl)
            tb = timeit(lambda: torch.autograd.grad(y, (x, fl), g, retain_graph=True))
            print('warp %s [%d,%d,%d,%d] %s fwd %7.1f us (%6.0f GB/s)   bwd %7.1f us (%6.0f GB/s)' % (
                name, B, C, h, w, kind, tf, fb / tf / 1e3, tb, bb / tb / 1e3), flush=True)
    for s in range(3):
        h, w = 256 >> s, 832 >> s
        x = torch.rand(8, 3, h, w, device='cuda')
        fl = _smooth_flow(8, h, w).requires_grad_()
        g = torch.randn(8, 3, h, w, device='cuda')
        tf = timeit(lambda: ops.warp_flow_masked(x, fl.detach()))
        y, _ = ops.warp_flow_masked(x, fl)
        tb = timeit(lambda: torch.autograd.grad(y, (fl,), g, retain_graph=True))
        fb, bb = 8 * h * w * 33, 4 * 8 * h * w * 10
        print('imgwarp s%d [8,3,%d,%d] fwd %7.1f us (%6.0f GB/s)   bwd %7.1f us (%6.0f GB/s)' % (
            s, h, w, tf, fb / tf / 1e3, tb, bb / tb / 1e3), flush=True)


def _sweep(envs, fn):
    """Run fn() under each environment override (tuning library only); restores the environment."""
    for env in envs:
        old = {k: os.environ.get(k) for k in env}
        os.environ.update({k: str(v) for k, v in env.items()})
        try:
            fn(' '.join('%s=%s' % (k.replace('UNFLOW_', ''), v) for k, v in env.items()) or 'default')
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v


def warp_c(B=16):
    """Feature warp through the C ABI (no autograd dispatch in the timings), tile-shape sweep on the tuning library."""
    lib = _lib.load()
    P = ops._ptr
    envs = [{}]
    if os.environ.get('UNFLOW_MICROBENCH_TUNING') == '1':
        envs = [{'UNFLOW_WARP_TILES': 0}] + [{'UNFLOW_WARP_BWD': k, 'UNFLOW_WARP_WGS': n, 'UNFLOW_WARP_MINPIX': 128} for k in (0, 1) for n in (256, 512, 1024)]
    for name, (C, h, w) in list(LEVELS.items())[:4]:
        x = torch.randn(B, C, h, w, device='cuda')
        g = torch.randn(B, C, h, w, device='cuda')
        out, gsrc = torch.empty_like(x), torch.empty_like(x)
        fb, bb = 4 * B * h * w * (2 * C + 2), 4 * B * h * w * (3 * C + 4)
        for kind, fl in (('smooth', _smooth_flow(B, h, w)), ('noise', torch.randn(B, 2, h, w, device='cuda') * 2)):
            gfl = torch.empty_like(fl)

            def run(tag):
                tf = timeit(lambda: lib.unflow_warp_fwd(P(x), P(fl), P(out), None, B, C, h, w, 0, ops._stream()))
                tb = timeit(lambda: lib.unflow_warp_bwd(P(x), P(fl), P(g), None, P(gsrc), P(gfl), B, C, h, w, 0, ops._stream()))
                print('warp_c %s [%d,%d,%d,%d] %-6s %-28s fwd %7.1f us (%6.0f GB/s)   bwd %7.1f us (%6.0f GB/s)' % (
                    name, B, C, h, w, kind, tag, tf, fb / tf / 1e3, tb, bb / tb / 1e3), flush=True)
            _sweep(envs, run)


def corr_bwd_sweep(B=16):
    """d=4 backward variants (tuning library): the default pick against the group-split ring kernel (4) and the tile kernel (6)."""
    lib = _lib.load()
    P = ops._ptr
    envs = [{}] + [{'UNFLOW_CORR_BWD': v, 'UNFLOW_CORR_GROUPS': g_} for v in (4, 6) for g_ in (1, 4)]
    for name, (C, h, w) in list(LEVELS.items())[:3]:
        f1 = torch.randn(B, C, h, w, device='cuda')
        f2 = torch.randn(B, C, h, w, device='cuda')
        g = torch.randn(B, 81, h, w, device='cuda')
        gf1, gf2 = torch.empty_like(f1), torch.empty_like(f2)
        bb = 4 * B * h * w * (4 * C + 81)
        ref = {}

        def run(tag):
            gf1.zero_(); gf2.zero_()
            tb = timeit(lambda: lib.unflow_corr_bwd(P(f1), P(f2), P(g), P(gf1), P(gf2), B, C, h, w, 4, ops._stream()))
            if not ref:
                ref['a'], ref['b'] = gf1.clone(), gf2.clone()
            err = max((gf1 - ref['a']).abs().max().item(), (gf2 - ref['b']).abs().max().item())
            print('corr_bwd %s [%d,%d,%d,%d] %-34s %7.1f us (%6.0f GB/s)  max|diff vs default| %.2e' % (
                name, B, C, h, w, tag, tb, bb / tb / 1e3, err), flush=True)
        _sweep(envs, run)


def warp_gather(B=16):
    """Round 3: feature-warp backward with the source gradient as a gather (default) against the scatter forms
    (UNFLOW_WARP_GATHER=0, tuning library), through the C entry point (no autograd dispatch in the timing); the largest
    difference between the two results."""
    lib = _lib.load()
    P = ops._ptr
    for name, (C, h, w) in list(LEVELS.items())[:3]:
        x = torch.randn(B, C, h, w, device='cuda')
        g = torch.randn(B, C, h, w, device='cuda')
        bb = 4 * B * h * w * (3 * C + 4)
        for kind, fl in (('smooth', _smooth_flow(B, h, w)), ('shift+smooth', _smooth_flow(B, h, w) + 5.3), ('noise', torch.randn(B, 2, h, w, device='cuda') * 2)):
            fl = fl.contiguous()
            gsrc, gflow = torch.empty_like(x), torch.empty_like(fl)
            ref = {}

            def run(tag):
                gsrc.fill_(float('nan')); gflow.fill_(float('nan'))
                tb = timeit(lambda: lib.unflow_warp_bwd(P(x), P(fl), P(g), None, P(gsrc), P(gflow), B, C, h, w, 0, ops._stream()))
                if not ref:
                    ref['a'], ref['b'] = gsrc.clone(), gflow.clone()
                err = max((gsrc - ref['a']).abs().max().item(), (gflow - ref['b']).abs().max().item())
                print('warp_bwd %s [%d,%d,%d,%d] %-12s %-28s %7.1f us (%6.0f GB/s)  max|diff vs first| %.2e (max|ref| %.2f)' % (
                    name, B, C, h, w, kind, tag, tb, bb / tb / 1e3, err, ref['a'].abs().max().item()), flush=True)
            _sweep([{}, {"UNFLOW_WARP_GATHER": 1}], run)


def corr_bwd_mf(B=16):
    """Round 5: the cost-volume backward on the matrix cores (csrc/corr_mfma.h) against the fp32 kernels at levels 2-4 of 832x256 and
    level 2 of 1024x448, d = 4 and 8; with the tuning library also the rows-per-wave sweep (UNFLOW_CORR_MF_ROWS)."""
    lib = _lib.load()
    P = ops._ptr
    tuning = os.environ.get('UNFLOW_MICROBENCH_TUNING') == '1'
    shapes = [('L2', 16, 32, 64, 208), ('L3', 16, 64, 32, 104), ('L4', 16, 96, 16, 52), ('S2', 8, 32, 112, 256), ('S3', 8, 64, 56, 128)]
    for d in (4, 8):
        D2 = (2 * d + 1) ** 2
        for name, Bq, C, h, w in shapes:
            f1 = torch.randn(Bq, C, h, w, device='cuda')
            f2 = torch.randn(Bq, C, h, w, device='cuda')
            g = torch.randn(Bq, D2, h, w, device='cuda') * 0.05
            gf1, gf2 = torch.empty_like(f1), torch.empty_like(f2)
            bb = 4 * Bq * h * w * (4 * C + D2)
            ref = {}

            def run(tag, mode):
                gf1.fill_(float('nan')); gf2.fill_(float('nan'))
                tb = timeit(lambda: lib.unflow_corr_bwd_ex(P(f1), P(f2), P(g), P(gf1), P(gf2), Bq, C, h, w, d, mode, ops._stream()), n=30)
                if not ref:
                    ref['a'], ref['b'] = gf1.clone(), gf2.clone()
                err = max((gf1 - ref['a']).abs().max().item(), (gf2 - ref['b']).abs().max().item())
                print('corr_bwd d=%d %s [%d,%d,%d,%d] %-22s %7.1f us (%6.0f GB/s = %.3f of 8 TB/s)  max|diff vs fp32| %.2e (max|ref| %.2f)' % (
                    d, name, Bq, C, h, w, tag, tb, bb / tb / 1e3, bb / tb / 1e3 / 8000.0, err, ref['a'].abs().max().item()), flush=True)
            run('fp32 kernels', 1)
            if tuning:
                _sweep([{'UNFLOW_CORR_MF_ROWS': r} for r in (8, 16, 32, 64)], lambda tag: run('mfma ' + tag, 2))
            else:
                run('mfma (shipped pick)', 2)


def corr_small(B=16):
    """Round 6: the small-map cost-volume backward with the gradient rows through registers (csrc/corr_small_rows.h, UNFLOW_CORR_BWD_FP32_NEXT = 3,
    never run on a GPU) against what unflow_corr_bwd runs today at levels 5 / 6 of 832x256 and 1024x448, d = 4 (corr_bwd_small_kernel) and d = 8
    (one lane per output element: 119 / 57 us in round 5)."""
    lib = _lib.load()
    P = ops._ptr
    shapes = [('L5', B, 128, 8, 26), ('L6', B, 196, 4, 13), ('S5', B // 2, 128, 14, 32), ('S6', B // 2, 196, 7, 16)]
    for d in (8, 4):
        D2 = (2 * d + 1) ** 2
        for name, Bq, C, h, w in shapes:
            f1 = torch.randn(Bq, C, h, w, device='cuda')
            f2 = torch.randn(Bq, C, h, w, device='cuda')
            g = torch.randn(Bq, D2, h, w, device='cuda') * 0.05
            gf1, gf2 = torch.empty_like(f1), torch.empty_like(f2)
            bb = 4 * Bq * h * w * (4 * C + D2)
            ref = {}
            for tag, mode in (('today (fp32)', 1), ('rows through registers', 3)):
                gf1.fill_(float('nan')); gf2.fill_(float('nan'))
                tb = timeit(lambda: lib.unflow_corr_bwd_ex(P(f1), P(f2), P(g), P(gf1), P(gf2), Bq, C, h, w, d, mode, ops._stream()), n=50)
                if not ref:
                    ref['a'], ref['b'] = gf1.clone(), gf2.clone()
                err = max((gf1 - ref['a']).abs().max().item(), (gf2 - ref['b']).abs().max().item())
                print('corr_bwd d=%d %s [%d,%d,%d,%d] %-24s %7.1f us (%6.0f GB/s = %.3f of 8 TB/s)  max|diff vs today| %.2e (max|ref| %.2f)' % (
                    d, name, Bq, C, h, w, tag, tb, bb / tb / 1e3, bb / tb / 1e3 / 8000.0, err, ref['a'].abs().max().item()), flush=True)


def ablate(B=16):
    """Phase ablations at level 2 (tuning library; results are wrong by construction, only the times matter)."""
    lib = _lib.load()
    P = ops._ptr
    C, h, w = LEVELS['L2']
    x = torch.randn(B, C, h, w, device='cuda'); g = torch.randn(B, C, h, w, device='cuda')
    gsrc = torch.empty_like(x); fl = _smooth_flow(B, h, w); gfl = torch.empty_like(fl)
    for kern in (0, 1):
        for dbg in ((0, 1, 2, 4, 3) if kern == 0 else (0, 1, 4, 8, 12)):
            os.environ.update({'UNFLOW_WARP_BWD': str(kern), 'UNFLOW_WARP_DEBUG': str(dbg)})
            tb = timeit(lambda: lib.unflow_warp_bwd(P(x), P(fl), P(g), None, P(gsrc), P(gfl), B, C, h, w, 0, ops._stream()))
            print('warp_bwd L2 %s dbg=%d (1 no LDS adds / no cell phase, 2 no flush, 4 no global atomics): %7.1f us' % (
                'cell-gather' if kern else 'LDS-accumulator', dbg, tb), flush=True)
    os.environ.pop('UNFLOW_WARP_DEBUG'); os.environ.pop('UNFLOW_WARP_BWD')
    tz = timeit(lambda: gsrc.zero_())
    print('zero-fill of gsrc alone: %.1f us' % tz, flush=True)
    f1 = torch.randn(B, C, h, w, device='cuda'); f2 = torch.randn(B, C, h, w, device='cuda')
    gc = torch.randn(B, 81, h, w, device='cuda'); gf1, gf2 = torch.empty_like(f1), torch.empty_like(f2)
    for fb in (4,):
        for dbg in (0, 8):
            os.environ.update({'UNFLOW_CORR_BWD': str(fb), 'UNFLOW_CORR_DEBUG': str(dbg)})
            tb = timeit(lambda: lib.unflow_corr_bwd(P(f1), P(f2), P(gc), P(gf1), P(gf2), B, C, h, w, 4, ops._stream()))
            print('corr_bwd L2 variant %d dbg=%d (8: no upstream-gradient gather): %7.1f us' % (fb, dbg, tb), flush=True)
    os.environ.pop('UNFLOW_CORR_BWD'); os.environ.pop('UNFLOW_CORR_DEBUG')


def pmc_l2(B=16):
    """The level-2 launches of corr / feature warp, a few times each: target of the PMC passes (tools/gpu_pmc.sh)."""
    lib = _lib.load()
    P = ops._ptr
    for lvl in ('L2', 'L3'):
        C, h, w = LEVELS[lvl]
        f1 = torch.randn(B, C, h, w, device='cuda'); f2 = torch.randn(B, C, h, w, device='cuda')
        gc = torch.randn(B, 81, h, w, device='cuda'); cv = torch.empty_like(gc)
        gf1, gf2 = torch.empty_like(f1), torch.empty_like(f2)
        fl = _smooth_flow(B, h, w); gfl = torch.empty_like(fl)
        for _ in range(3):
            lib.unflow_corr_fwd(P(f1), P(f2), P(cv), B, C, h, w, 4, ops._stream())
            lib.unflow_corr_bwd(P(f1), P(f2), P(gc), P(gf1), P(gf2), B, C, h, w, 4, ops._stream())
            lib.unflow_warp_fwd(P(f1), P(fl), P(gf1), None, B, C, h, w, 0, ops._stream())
            lib.unflow_warp_bwd(P(f1), P(fl), P(f2), None, P(gf1), P(gfl), B, C, h, w, 0, ops._stream())
        torch.cuda.synchronize()


def corr_split_sweep(B=16):
    """Channel slices of the small-map forward (tuning library): 1 = per-element kernel."""
    lib = _lib.load()
    P = ops._ptr
    for lvl, d in (('L5', 4), ('L6', 4), ('L5', 8), ('L6', 8)):
        C, h, w = LEVELS[lvl]
        f1 = torch.randn(B, C, h, w, device='cuda'); f2 = torch.randn(B, C, h, w, device='cuda')
        cv = torch.empty(B, (2 * d + 1) ** 2, h, w, device='cuda')

        def run(tag):
            t = timeit(lambda: lib.unflow_corr_fwd(P(f1), P(f2), P(cv), B, C, h, w, d, ops._stream()))
            print('corr_fwd d=%d %s %-18s %6.1f us' % (d, lvl, tag, t), flush=True)
        _sweep([{'UNFLOW_CORR_SPLIT': k} for k in (1, 2, 4, 8)], run)


def corr_fwd_sweep(B=16):
    """d=4 forward variants (tuning library) at levels 2-4."""
    lib = _lib.load()
    P = ops._ptr
    envs = [{}] + [{'UNFLOW_CORR_VARIANT': v} for v in (7, 9, 12, 13)]
    for name, (C, h, w) in list(LEVELS.items())[:3]:
        f1 = torch.randn(B, C, h, w, device='cuda'); f2 = torch.randn(B, C, h, w, device='cuda')
        cv = torch.empty(B, 81, h, w, device='cuda')
        fb = 4 * B * h * w * (2 * C + 81)
        ref = {}

        def run(tag):
            cv.zero_()
            tf = timeit(lambda: lib.unflow_corr_fwd(P(f1), P(f2), P(cv), B, C, h, w, 4, ops._stream()))
            if not ref:
                ref['a'] = cv.clone()
            print('corr_fwd %s [%d,%d,%d,%d] %-24s %7.1f us (%6.0f GB/s)  max|diff vs default| %.2e' % (
                name, B, C, h, w, tag, tf, fb / tf / 1e3, (cv - ref['a']).abs().max().item()), flush=True)
        _sweep(envs, run)


def fused(B=16):
    """Fused warp + cost volume forward against the two separate launches (C ABI)."""
    lib = _lib.load()
    P = ops._ptr
    for name, (C, h, w) in list(LEVELS.items())[:4]:
        f1 = torch.randn(B, C, h, w, device='cuda'); f2 = torch.randn(B, C, h, w, device='cuda')
        cv = torch.empty(B, 81, h, w, device='cuda'); wp = torch.empty_like(f2)
        for kind, fl in (('smooth', _smooth_flow(B, h, w)), ('noise', torch.randn(B, 2, h, w, device='cuda') * 2)):
            tfu = timeit(lambda: lib.unflow_warp_corr_fwd(P(f1), P(f2), P(fl), P(cv), B, C, h, w, 4, 0, ops._stream()))
            a = cv.clone()

            def two():
                lib.unflow_warp_fwd(P(f2), P(fl), P(wp), None, B, C, h, w, 0, ops._stream())
                lib.unflow_corr_fwd(P(f1), P(wp), P(cv), B, C, h, w, 4, ops._stream())
            tsep = timeit(two)
            nb = 4 * B * h * w * (4 * C + 2 + 81)
            print('fused %s [%d,%d,%d,%d] %-6s fused %7.1f us (%6.0f GB/s)   separate %7.1f us   max|diff| %.2e' % (
                name, B, C, h, w, kind, tfu, nb / tfu / 1e3, tsep, (a - cv).abs().max().item()), flush=True)


def _stamp_report(tag, buf, names):
    t = buf.view(1024, 3, 8).cpu()
    seg = t[:, :, :len(names)].double()
    first, last = t[:, :, 6].double(), t[:, :, 7].double()
    used = seg.sum(2) > 0
    t0 = first[used].min()
    for grp in range(3):
        u = used[:, grp]
        if not u.any():
            continue
        m = seg[:, grp][u]
        med = m.median(0).values
        st, en = first[:, grp][u] - t0, last[:, grp][u] - t0
        print('%s wave group %d (%d workgroups): %6.0f ticks | ' % (tag, grp, m.shape[0], med.sum()) +
              '  '.join('%s %4.1f%%' % (n, 100 * v / med.sum()) for n, v in zip(names, med.tolist())) +
              ' | starts %.0f..%.0f, ends %.0f..%.0f (median %.0f)' % (st.min(), st.max(), en.min(), en.max(), en.median()), flush=True)


def stamps(B=16):
    """In-kernel s_memtime stamps (tuning library): where a wave's cycles go in the group-split cost-volume backward and
    in the ring forward, and when workgroups start / end."""
    lib = _lib.load()
    P = ops._ptr
    bwd_names = ['gather+prologue', 'vmcnt wait + barrier', 'finish (grp 0)', 'DMA issue', 'rows + hand-off', 'tail']
    fwd_names = ['set-up+prologue', 'vmcnt wait + barrier', 'DMA issue', 'row pipeline', 'store issue']
    for lvl in ('L2', 'L3', 'L4'):
        C, h, w = LEVELS[lvl]
        f1 = torch.randn(B, C, h, w, device='cuda'); f2 = torch.randn(B, C, h, w, device='cuda')
        gc = torch.randn(B, 81, h, w, device='cuda'); gf1, gf2 = torch.empty_like(f1), torch.empty_like(f2)
        cv = torch.empty_like(gc)
        calls = {'bwd': lambda: lib.unflow_corr_bwd(P(f1), P(f2), P(gc), P(gf1), P(gf2), B, C, h, w, 4, ops._stream())}
        if lvl == 'L2':
            calls['fwd'] = lambda: lib.unflow_corr_fwd(P(f1), P(f2), P(cv), B, C, h, w, 4, ops._stream())
        for kind, fn in calls.items():
            buf = torch.zeros(1024 * 3 * 8, dtype=torch.int64, device='cuda')
            us = timeit(fn)
            print('%s %s  UNFLOW_CORR_BWD=%s  %.1f us' % (lvl, kind, os.environ.get('UNFLOW_CORR_BWD', '-'), us), flush=True)
            os.environ['UNFLOW_STAMP_PTR'] = str(buf.data_ptr())
            fn()
            torch.cuda.synchronize()
            os.environ.pop('UNFLOW_STAMP_PTR')
            _stamp_report('%s %s' % (lvl, kind), buf, bwd_names if kind == 'bwd' else fwd_names)


def epilogue(B=16):
    """conv() epilogue (bias + LeakyReLU, in place) and its backward on NCHW and channels_last activations."""
    for (C, h, w, n) in ((128, 64, 208, B), (32, 64, 208, B), (96, 16, 52, B), (16, 128, 416, 24)):
        for cl in (False, True):
            y = torch.randn(n, C, h, w, device='cuda')
            g = torch.randn(n, C, h, w, device='cuda')
            if cl:
                y, g = y.contiguous(memory_format=torch.channels_last), g.contiguous(memory_format=torch.channels_last)
            bias = torch.randn(C, device='cuda', requires_grad=True)
            tf = timeit(lambda: ops.bias_leaky_relu_(y, bias.detach(), 0.1))
            yy = y.clone().requires_grad_()
            out = ops.bias_leaky_relu_(yy * 1.0, bias, 0.1)
            tb = timeit(lambda: torch.autograd.grad(out, (yy, bias), g, retain_graph=True))
            nb = y.numel() * 4
            print('epilogue [%d,%d,%d,%d] %-13s fwd %6.1f us (%5.0f GB/s)   bwd (incl. the x*1.0 node) %6.1f us (%5.0f GB/s)' % (
                n, C, h, w, 'channels_last' if cl else 'NCHW', tf, 2 * nb / tf / 1e3, tb, 3 * nb / tb / 1e3), flush=True)


def layout(B=16):
    """Layout glue of the channels_last conv stacks: cat -> NHWC (decoder input) and NHWC -> NCHW (pyramid features), against torch."""
    CL = torch.channels_last
    for (chans, h, w, n) in (((81, 32, 2), 64, 208, B), ((81, 64, 2), 32, 104, B), ((32,), 64, 208, 24)):
        xs = [torch.randn(n, c, h, w, device='cuda') for c in chans]
        nb = 8 * n * sum(chans) * h * w
        t_k = timeit(lambda: ops.cat_channels_last(xs))
        t_t = timeit(lambda: torch.cat(xs, 1).contiguous(memory_format=CL))
        y = ops.cat_channels_last(xs)
        t_b = timeit(lambda: ops.to_nchw(y))
        t_bt = timeit(lambda: y.contiguous())
        print('layout [%d,%s,%d,%d]  cat->NHWC %6.1f us (%5.0f GB/s; torch cat + contiguous %6.1f)   NHWC->NCHW %6.1f us (%5.0f GB/s; torch %6.1f)' % (
            n, '+'.join(map(str, chans)), h, w, t_k, nb / t_k / 1e3, t_t, t_b, nb / t_b / 1e3, t_bt), flush=True)


def losses(B=8):
    for s in range(3):
        h, w = 256 >> s, 832 >> s
        img = torch.rand(B, 3, h, w, device='cuda')
        wp = torch.rand(B, 3, h, w, device='cuda').requires_grad_()
        wp2 = torch.rand(B, 3, h, w, device='cuda').requires_grad_()
        wt = torch.rand(B, 1, h, w, device='cuda')
        fl = (torch.randn(B, 2, h, w, device='cuda') * 3).requires_grad_()
        fb = torch.randn(B, 2, h, w, device='cuda') * 3
        gl = torch.ones(B, device='cuda')
        n = B * h * w
        for nm, fwd, inp, bytes_f, bytes_b in (
                ('ssim', lambda: ops.ssim_loss(img, wp, wt), (wp,), 4 * n * 7, 4 * n * 10),
                ('occ_weight', lambda: ops.occ_weight(img, wp, wp2)[0].sum((1, 2, 3)), (wp,), 4 * n * 13 + 2 * n, 4 * n * 10),
                ('smooth2', lambda: ops.smooth2_loss(fl, img), (fl,), 4 * n * 5, 4 * n * 7),
                ('consis', lambda: ops.consis_loss(fl, fb, wt), (fl,), 4 * n * 5, 4 * n * 7),
                ('masked_mean', lambda: ops.masked_mean(wt.requires_grad_(), wt.detach()), (wt,), 4 * n * 2, 4 * n * 2)):
            tf = timeit(fwd)
            y = fwd()
            tb = timeit(lambda: torch.autograd.grad(y, inp, gl, retain_graph=True))
            print('%-11s s%d [%d,.,%d,%d] fwd %7.1f us (%6.0f GB/s)   bwd %7.1f us (%6.0f GB/s)' % (
                nm, s, B, h, w, tf, bytes_f / tf / 1e3, tb, bytes_b / tb / 1e3), flush=True)


def prepare(B=8):
    """Input stage: B decoded KITTI triplets (uint8, 1242x375 x 3 frames) -> [B,3,768,832] fp32."""
    import ctypes
    import numpy as np
    for (h, w), (H, W) in (((375, 1242), (256, 832)), ((436, 1024), (448, 1024))):
        n_src = 3 * h * w * 3
        pad = (n_src + 15) // 16 * 16
        src = torch.randint(0, 256, (B * pad,), dtype=torch.uint8, device='cuda')
        offs = torch.tensor([i * pad for i in range(B)], dtype=torch.int64, device='cuda')
        dims = torch.tensor([[3 * h, w]] * B, dtype=torch.int32, device='cuda')
        out = torch.empty(B, 3, 3 * H, W, device='cuda')
        lib = _lib.load()
        P = ctypes.c_void_p
        st = P(torch.cuda.current_stream().cuda_stream)
        t = timeit(lambda: lib.unflow_prepare_triplets(P(src.data_ptr()), P(offs.data_ptr()), P(dims.data_ptr()), P(0),
                                                       P(out.data_ptr()), B, H, W, 1, st))
        nbytes = B * n_src + out.numel() * 4
        print('prepare     [%d x %dx%dx3 u8] -> [%d,3,%d,%d] f32  %7.1f us (%6.0f GB/s algorithmic)' % (
            B, 3 * h, w, B, 3 * H, W, t, nbytes / t / 1e3), flush=True)
        imgs = [np.random.randint(0, 256, (3 * h, w, 3), dtype=np.uint8) for _ in range(B)]
        staging = torch.empty(B * pad + 4096, dtype=torch.uint8).pin_memory()
        ops.prepare_triplets(imgs, (H, W), None, 'cuda:0', True, staging)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            ops.prepare_triplets(imgs, (H, W), None, 'cuda:0', True, staging)
        torch.cuda.synchronize()
        print('  host-inclusive (pack + pinned H2D + kernel): %.2f ms per batch of %d' % ((time.perf_counter() - t0) * 100, B), flush=True)


def loader(n_files=64, workers=8, B=8):
    """Input stage end to end: stacked KITTI-sized PNG triplets on disk -> DeviceTripletLoader batches
    (PNG inflate + unfilter on `workers` CPU processes, everything else in unflow_prepare_triplets)."""
    import tempfile
    import numpy as np
    from unopticalflow_amd.data import DecodedTriplets, DeviceTripletLoader, PreparedTriplets
    from unopticalflow_amd.evaluation import write_png
    root = tempfile.mkdtemp(prefix='unflow_loader_')
    os.makedirs(os.path.join(root, 'seq'))
    rng = np.random.default_rng(0)
    yy, xx = np.mgrid[0:1125, 0:1242]
    names = []
    for i in range(n_files):            # smooth texture + noise: compresses roughly like a photograph (~1.7x)
        img = np.stack([127 + 90 * np.sin(xx / (17.0 + c + i % 5) + i) * np.cos(yy / (23.0 + c)) for c in range(3)], -1)
        img = np.clip(img + rng.normal(0, 6, img.shape), 0, 255).astype(np.uint8)
        write_png(os.path.join(root, 'seq', '%d.png' % i), img)
        names.append('seq/%d.png seq/%d_cam.txt' % (i, i))
    open(os.path.join(root, 'train.txt'), 'w').write('\n'.join(names) + '\n')
    size = sum(os.path.getsize(os.path.join(root, 'seq', f)) for f in os.listdir(os.path.join(root, 'seq'))) / n_files / 1e6
    for kind in ('device', 'host'):
        if kind == 'device':
            it = DeviceTripletLoader(DecodedTriplets(root, img_hw=(256, 832), num_iterations=40 * B), B, 'cuda:0', (256, 832), workers)
        else:
            ds = PreparedTriplets(root, img_hw=(256, 832), num_iterations=40 * B)
            dl = torch.utils.data.DataLoader(ds, batch_size=B, num_workers=workers, pin_memory=True)
            it = (b.cuda(non_blocking=True) for b in dl)
        n, t0 = 0, None
        for b in it:
            if t0 is None:              # first batch pays the worker start-up
                torch.cuda.synchronize(); t0 = time.perf_counter(); continue
            n += b.shape[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print('loader %-6s %d workers, %.1f MB PNGs: %.0f triplets/s (%.0f pairs/s)' % (kind, workers, size, n / dt, 2 * n / dt), flush=True)


if __name__ == '__main__':
    which = sys.argv[1:] or ['corr', 'warp', 'losses', 'prepare']
    for w_ in which:
        globals()[w_]()
