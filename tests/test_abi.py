"""CPU-side checks of the drop-in boundary: the C-ABI library builds for gfx950, loads, exports every
symbol include/unflow_hip.h declares with the arity the ctypes binding uses, and the product has no
CPU fallback.  No compute is launched (there is no GPU here)."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_decls():
    src = open(os.path.join(ROOT, 'include', 'unflow_hip.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    decls = {}
    for m in re.finditer(r'\bint\s+(unflow_\w+)\s*\(([^)]*)\)\s*;', src):
        args = m.group(2).strip()
        decls[m.group(1)] = 0 if args in ('', 'void') else len(args.split(','))
    return decls


def test_library_builds_loads_and_exports_every_header_symbol():
    import __graft_entry__ as ge
    ge.build()
    from unopticalflow_amd import _lib
    lib = _lib.load()
    decls = _header_decls()
    assert len(decls) >= 17
    assert set(decls) == set(_lib.SIGNATURES), set(decls) ^ set(_lib.SIGNATURES)
    for name, nargs in decls.items():
        assert hasattr(lib, name), name
        assert len(_lib.SIGNATURES[name]) == nargs, name
    assert lib.unflow_abi_version() == _lib.ABI_VERSION


def test_variant_libraries_in_the_tree_are_not_stale():
    """Variant builds (libunflow_hip_tuning*.so: tools/ and the gpu_r6.sh A/B recipes load them through UNFLOW_LIB_PATH) travel to the GPU
    box with the tree: one that was built before the header grew would fail there, not here.  Every one present exports every symbol
    the header declares and reports the header's ABI version."""
    pkg = os.path.join(ROOT, 'unopticalflow_amd')
    from unopticalflow_amd import _lib
    decls = _header_decls()
    for f in sorted(os.listdir(pkg)):
        if f.startswith('libunflow_hip_tuning') and f.endswith('.so'):
            lib = ctypes.CDLL(os.path.join(pkg, f))
            missing = [n for n in decls if not hasattr(lib, n)]
            assert not missing, (f, missing)
            assert lib.unflow_abi_version() == _lib.ABI_VERSION, f


def test_argument_validation_without_gpu():
    from unopticalflow_amd import _lib
    lib = _lib.load()
    null = ctypes.c_void_p(0)
    assert lib.unflow_corr_fwd(null, null, null, 1, 1, 1, 1, 4, null) == -22
    assert lib.unflow_warp_fwd(null, null, null, null, 1, 3, 8, 8, 0, null) == -22
    assert lib.unflow_partials_per_sample(0, 5) == -22
    assert lib.unflow_partials_per_sample(256, 832) > 0


def test_no_cpu_fallback():
    from unopticalflow_amd import ops, get_model
    from oracle import ref_cpu as R
    with pytest.raises(RuntimeError):
        ops.corr(torch.zeros(1, 2, 4, 4), torch.zeros(1, 2, 4, 4))
    with pytest.raises(RuntimeError):
        ops.warp_flow(torch.zeros(1, 3, 8, 8), torch.zeros(1, 2, 8, 8))
    model = get_model('flow')(R.default_cfg())
    with pytest.raises(RuntimeError):
        model(torch.rand(1, 3, 192, 64))


def test_product_does_not_import_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, 'unopticalflow_amd')
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                text = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle', text, flags=re.M), os.path.join(dp, f)


def test_get_model_unknown_mode():
    from unopticalflow_amd import get_model
    with pytest.raises(ValueError):
        get_model('depth')


def test_dma_ring_kernels_do_not_spill():
    """The LDS-DMA ring kernels wait with counted s_waitcnt vmcnt(N); scratch (spill) traffic shares that
    counter, so a spilling build would under-wait.  Guard: those kernels must compile without scratch."""
    import sys
    from unopticalflow_amd import build as b
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        import isa_hashes
    finally:
        sys.path.pop(0)
    # (the build's flags, one compile per file and process: shared with the instruction-stream hashes of the tests below)
    table = isa_hashes.isa_scratch(os.path.join(b.CSRC, 'corr.hip'))
    names, scratch = list(table), list(table.values())
    assert names
    checked = 0
    for n, s in zip(names, scratch):
        if 'ring_kernel' in n or 'ring_mixed_kernel' in n or 'gs_kernel' in n or 'rs_kernel' in n or 'rs_mixed_kernel' in n or 'mf_kernel' in n:      # (row-streamed backward: counted vmcnt behind its LDS-DMA too; matrix-core backward: two request sets in flight)
            assert s == 0, (n, s)
            checked += 1
    assert checked >= 3      # ring<9 rows>, ring<3 rows>, group-split backward


# Device code that differs from commit 8bbc033, whose `-m gpu` suite the driver ran green on an MI355X (GPUTEST_r04: 249 passed).
# Every other kernel the library ships is, instruction for instruction, one that suite exercised.  What is listed here has its own
# GPU evidence from round 5: see DESIGN.md section 7.
ROUND5_DEVICE_CODE = {
    'corr.hip': {'new': {'corr_bwd_mf_kernel<4, 2, 1, 1>', 'corr_bwd_mf_kernel<8, 2, 2, 1>',         # csrc/corr_mfma.h (on request: UNFLOW_CORR_BWD_MFMA)
                         # round 6, csrc/corr_small_rows.h: never run on a GPU, UNFLOW_CORR_BWD_FP32_NEXT only
                         'corr_bwd_smallrows_kernel<4, 3, 4>', 'corr_bwd_smallrows_kernel<4, 3, 8>', 'corr_bwd_smallrows_kernel<8, 3, 4>', 'corr_bwd_smallrows_kernel<8, 3, 8>'},
                 'renamed': {'corr_bwd_rs_kernel<4, 16, 8, 2>': 'corr_bwd_rs_kernel<4, 16, 8, 2, 0, 1, 1>',
                             'corr_bwd_rs_kernel<4, 8, 8, 2>': 'corr_bwd_rs_kernel<4, 8, 8, 2, 0, 1, 1>',
                             'corr_bwd_rs_kernel<8, 8, 8, 1>': 'corr_bwd_rs_kernel<8, 8, 8, 1, 0, 1, 1>'},
                 'removed': {'corr_bwd_gs_kernel<4, 2, 4>', 'corr_bwd_gs_kernel<4, 2, 8>', 'corr_bwd_gs_kernel<4, 4, 4>'}},
    'photo.hip': {'new': {'loss_finalize_batch_kernel',                                               # the batched second stage (GPU: smoke + bench)
                          # csrc/multiscale.h, never run: the single-scale kernels' bodies (csrc/bodies/*.inc) behind a table of scales; off by default
                          'occ_weight_fwd_ms_kernel', 'absdiff_bwd_ms_kernel', 'masked_mean_partial_ms_kernel', 'masked_mean_bwd_ms_kernel',
                          'smooth2_fwd_tile_ms_kernel', 'smooth2_bwd_stage_ms_kernel', 'consis_partial_ms_kernel', 'consis_bwd_ms_kernel'}},
    'warp.hip': {'new': {'warp_fwd_ms_kernel', 'warp_bwd_ms_kernel'}},                                # the masked image warps of a pyramid; never run, off by default
    'ssim.hip': {'new': {'ssim2_fwd_ms_kernel<false>', 'ssim2_bwd_ms_kernel'}},     # (round 5's fma forms of pair_factors were taken back in round 6: the six ssim2 kernels hash to round 4's)
}


def test_fp32_cost_volume_kernels_are_the_validated_ones():
    """The round-5 pruning of csrc/corr.hip (and everything else that was not meant to change device code) changed no machine
    code: per kernel, hipcc's gfx950 instruction stream hashes to what the round-4 sources give (tests/golden/isa_validated_r4.json,
    written by `tools/isa_hashes.py --tree`); the kernels that do differ are exactly ROUND5_DEVICE_CODE."""
    import json
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        import isa_hashes
    finally:
        sys.path.pop(0)
    golden = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'isa_validated_r4.json')))
    if isa_hashes.hipcc_version() != golden['hipcc']:
        pytest.skip('hashes were taken with %s, this is %s' % (golden['hipcc'], isa_hashes.hipcc_version()))
    now = isa_hashes.tree_hashes(ROOT)
    assert set(now) == set(golden['files'])
    for f, kernels in now.items():
        allow = ROUND5_DEVICE_CODE.get(f, {})
        old = golden['files'][f]
        renamed = allow.get('renamed', {})
        for k, v in kernels.items():
            if k in allow.get('new', ()):
                assert k not in old, (f, k)
                continue
            was = old.get(renamed.get(k, k))
            assert was is not None, 'kernel %s of %s did not exist in round 4 and is not listed' % (k, f)
            if k in allow.get('changed', ()):
                assert was['sha16'] != v['sha16'], 'listed as changed but identical: ' + k
            else:
                assert was == v, 'device code of %s (%s) differs from the GPU-validated build' % (k, f)
        gone = set(old) - set(kernels) - set(renamed.values())
        assert gone == allow.get('removed', set()), (f, gone)


def test_round5_host_entry_points_without_gpu():
    """The host-only entry points of ABI 10-12: the per-call backward arithmetic of the cost volume rejects unknown values (and the process-wide
    setter of ABI 10-11 is gone); the
    partial-sum counts that ``unflow_loss_finalize_batch`` jobs are described with follow the kernels' tilings (masked mean /
    consistency: 2048 pixels per workgroup; smoothness: 64 x 8 tiles; SSIM loss: 124-column strips x 8- or 16-row chunks when every
    tensor is 8-byte aligned and the width even) and never exceed the scratch ``unflow_partials_per_sample`` sizes."""
    from unopticalflow_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    assert not hasattr(lib, 'unflow_corr_set_backward')          # ABI 12: the arithmetic is an argument of unflow_corr_bwd_ex, the library keeps no mode
    one = ctypes.c_void_p(16)                                      # (argument checks come before any launch: a non-NULL dummy pointer is never touched)
    lib.unflow_corr_bwd_ex.argtypes = _lib.SIGNATURES['unflow_corr_bwd_ex']
    for bad in (-1, 4, 7):
        assert lib.unflow_corr_bwd_ex(one, one, one, one, one, 1, 16, 8, 8, 4, bad, None) == -22
    assert lib.unflow_corr_bwd_ex(None, one, one, one, one, 1, 16, 8, 8, 4, 0, None) == -22
    cdiv = lambda a, b: (a + b - 1) // b
    for H, W in ((256, 832), (128, 416), (64, 208), (33, 57), (448, 1024), (1, 1)):
        per_sample = lib.unflow_partials_per_sample(H, W) // 2
        assert lib.unflow_loss_partial_blocks(0, H, W, 16, 1) == lib.unflow_loss_partial_blocks(3, H, W, 16, 1) == cdiv(H * W, 2048)
        assert lib.unflow_loss_partial_blocks(2, H, W, 16, 1) == cdiv(W, 64) * cdiv(H, 8)
        fast = lib.unflow_loss_partial_blocks(1, H, W, 16, 1)
        if W % 2 == 0:
            assert fast == cdiv(W, 124) * cdiv(H, 16 if H >= 128 else 8)
        else:
            assert fast == lib.unflow_loss_partial_blocks(1, H, W, 16, 0)
        for op in range(4):
            for aligned in (0, 1):
                assert 0 < lib.unflow_loss_partial_blocks(op, H, W, 16, aligned) <= per_sample
    assert lib.unflow_loss_partial_blocks(4, 8, 8, 1, 1) == -22 and lib.unflow_loss_partial_blocks(0, 0, 8, 1, 1) == -22


def test_multiscale_entries_reject_what_they_do_not_serve():
    """ABI 11 (`_ms`: one launch over the scales of a loss, csrc/multiscale.h): every argument check comes before the launch, so it can be
    exercised here -- no scales, more than four, a NULL table, a NULL entry, B < 1, and the shapes the single-scale entries serve through
    another kernel (odd width / unaligned tensor in the SSIM pair, a map under 3 x 3 in the smoothness backward, B > 65535)."""
    from unopticalflow_amd import _lib
    lib = _lib.load()
    ptrs = lambda *v: (ctypes.c_void_p * len(v))(*v)
    ints = lambda *v: (ctypes.c_int * len(v))(*v)
    ok = 4096                                                          # (a fake 8-byte aligned device address: never dereferenced before the checks)
    for n in (0, 5, -1):
        assert lib.unflow_masked_mean_fwd_ms(n, ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(8), 2, None) == -22
        assert lib.unflow_ssim_loss_fwd_ms(n, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(8), 2, 1, None) == -22
    assert lib.unflow_occ_weight_fwd_ms(1, None, ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(8), 2, None) == -22
    assert lib.unflow_occ_weight_fwd_ms(1, ptrs(ok), ptrs(0), ptrs(ok), ptrs(ok), ints(8), ints(8), 2, None) == -22
    assert lib.unflow_occ_weight_fwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(8), 0, None) == -22
    assert lib.unflow_absdiff_bwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(8), 3, 2, None) == -22       # B % img_batch
    assert lib.unflow_masked_mean_bwd_ms(2, ptrs(ok, ok), ptrs(ok, 0), ptrs(ok, ok), ptrs(ok, ok), ints(8, 4), ints(8, 4), 2, None) == -22
    assert lib.unflow_ssim_loss_fwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(9), 2, 1, None) == -22      # odd width
    assert lib.unflow_ssim_loss_fwd_ms(1, ptrs(ok), ptrs(ok + 4), ptrs(ok), ptrs(ok), ints(8), ints(8), 2, 1, None) == -22  # 4-byte aligned
    assert lib.unflow_ssim_loss_bwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok + 4), ints(8), ints(8), 2, 1, None) == -22
    assert lib.unflow_smooth2_bwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ints(2), ints(8), 2, 1, None) == -22       # H < 3
    assert lib.unflow_smooth2_fwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(8), 65536, 65536, None) == -22
    assert lib.unflow_consis_fwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(0), 2, None) == -22
    assert lib.unflow_consis_bwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(8), 0, None) == -22
    assert lib.unflow_warp_fwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(ok), ints(8), ints(8), 2, 5, 0, None) == -22          # C > 4: feature maps go per level
    assert lib.unflow_warp_fwd_ms(1, ptrs(ok), ptrs(ok), ptrs(ok), ptrs(0), ints(8), ints(8), 2, 3, 0, None) == -22           # the mask is not optional here
    assert lib.unflow_warp_bwd_ms(2, ptrs(ok, ok), ptrs(ok, ok), ptrs(ok, ok), ptrs(ok, ok), ptrs(ok, 0), ints(8, 4), ints(8, 4), 2, 3, 0, None) == -22


def test_no_wide_buffer_store_names_a_scalar_offset():
    """hipcc (ROCm 7.2) skips the "VMEM store of more than 64 bits -> VALU write of the store data" hazard when the store's soffset is an
    SGPR, and gfx950 then stores what the following instruction puts into those registers (found the hard way in round 5:
    profiles/r5_corr_bwd_mfma.md, finding 1; the matrix-core kernel folds its channel-group offset into the vector offset since).
    Guard for every kernel the library ships: no buffer_store_dwordx3 / x4 with a register soffset."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        import isa_hashes
    finally:
        sys.path.pop(0)
    csrc = os.path.join(ROOT, 'unopticalflow_amd', 'csrc')
    wide = bad = 0
    for f in sorted(os.listdir(csrc)):
        if not f.endswith('.hip'):
            continue
        for k, ins in isa_hashes.isa_streams(os.path.join(csrc, f)).items():
            for line in ins:
                m = re.match(r'buffer_store_dwordx[34]\s+(\S+),\s*(\S+),\s*(s\[\d+:\d+\]),\s*(\S+)', line)
                if m:
                    wide += 1
                    if re.match(r's\d+|s\[', m.group(4)):
                        bad += 1
                        print(f, k, line)
    assert wide > 50 and bad == 0, (wide, bad)


def test_matrix_core_backward_waits_are_counted():
    """The second hipcc finding of round 5 (profiles/r5_corr_bwd_mfma.md): the s_waitcnt pass takes the weaker of the loop pre-header's
    and the back edge's view of what is in flight, and an unpinned request group turns every first use in the loop into `vmcnt(0)` -- a
    full drain of the loads the design keeps in flight.  Guard: in corr_bwd_mf_kernel no loop that issues MFMAs waits for vmcnt(0)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    try:
        import isa_hashes
    finally:
        sys.path.pop(0)
    st = isa_hashes.isa_streams(os.path.join(ROOT, 'unopticalflow_amd', 'csrc', 'corr.hip'))
    names = isa_hashes.demangle(list(st))
    checked = 0
    for k, ins in st.items():
        if 'corr_bwd_mf_kernel' not in names[k]:
            continue
        labels = {l[:-1]: i for i, l in enumerate(ins) if l.endswith(':')}
        for i, l in enumerate(ins):
            m = re.match(r's_c?branch\w*\s+(\.LBB_\d+)', l)
            if m and labels.get(m.group(1), i) < i:                          # a back edge: ins[target:i] is a loop body
                body = ins[labels[m.group(1)]:i]
                if any('v_mfma' in x for x in body):
                    checked += 1
                    assert not [x for x in body if x.startswith('s_waitcnt') and 'vmcnt(0)' in x], (names[k], labels[m.group(1)], i)
    assert checked >= 8


def test_pmc_traffic_carries_over_to_byte_identical_kernels(tmp_path):
    """bench.py's `roofline.traffic` comes from a committed rocprofv3 --pmc capture.  The capture names the kernels that served each entry
    and the hash of each one's instruction stream; a later build is served by it exactly when hipcc still emits those kernels byte for
    byte (the round-5 sources differ from the capture's in comments and pruned templates only: the level-2 cost-volume backward keeps its
    measured 211.3 MB, the SSIM loss its 113.2 MB) -- and not when a hash in the file does not match."""
    import json
    import bench
    got, note = bench.measured_traffic('unflow_corr_bwd', [16, 32, 64, 208])
    assert got == 211299925 and 'byte-identical' in note and 'corr_bwd_rs_mixed_kernel<4, 16, 2>' in note, (got, note)
    got, note = bench.measured_traffic('unflow_ssim_loss_fwd', [16, 3, 256, 832])          # (round 6 took the SSIM kernels back to round 4's machine code)
    assert got == 113197141 and 'byte-identical' in note and 'ssim2_fwd_kernel<16, false>' in note, (got, note)
    d = json.load(open(os.path.join(ROOT, 'profiles', 'r4_pmc_traffic.json')))
    d['kernel_isa']['kernels']['corr_bwd_rs_mixed_kernel<4, 16, 2>']['sha16'] = '0' * 16
    p = tmp_path / 'tampered.json'
    p.write_text(json.dumps(d))
    got, note = bench.measured_traffic('unflow_corr_bwd', [16, 32, 64, 208], path=str(p))
    assert got is None and 'differs' in note


def test_every_shared_body_is_included_by_a_single_scale_and_a_multi_scale_kernel():
    """The `_ms` kernels' claim to the single-scale kernels' bits rests on both including the SAME body file (csrc/bodies/*.inc) and on
    the single-scale kernel being nothing but that body: every body file is included at least twice, once by a kernel whose whole
    definition is the include, and the other includers are `_ms` kernels that open with UNFLOW_MS_PROLOGUE."""
    csrc = os.path.join(ROOT, 'unopticalflow_amd', 'csrc')
    bodies = sorted(f for f in os.listdir(os.path.join(csrc, 'bodies')) if f.endswith('.inc'))
    assert len(bodies) == 12
    text = ''.join(open(os.path.join(csrc, f)).read() for f in ('photo.hip', 'ssim.hip', 'warp.hip', 'ms_flat_photo.h', 'ms_flat_warp.h'))
    for b in bodies:
        inc = '#include "bodies/%s"' % b
        assert text.count(inc) >= 2, b
        assert re.search(r'\) \{\n' + re.escape(inc) + r'\n\}\n', text), 'no kernel that is only ' + inc         # the single-scale kernel
        ms = re.findall(r'void (\w+_ms_kernel)\(MsTable<\w+> ms_table_[^)]*\) \{\n    UNFLOW_MS_PROLOGUE\(ms_table_\);(?:(?!\n\}\n).)*?' + re.escape(inc), text, flags=re.S)
        assert len(ms) >= 1, 'no _ms kernel includes ' + b


def test_multiscale_workgroup_table_covers_every_scale_once():
    """csrc/multiscale.h restated: ms_grid_add pads a scale's workgroup range to a multiple of 8 and ms_locate maps a linear workgroup id
    to (scale, virtual blockIdx).  For the grids of the train step's ten loss kernels: every (scale, x, y, z) of the per-scale grids is
    produced exactly once, padding workgroups are recognised, a scale starts on an XCD boundary (id mod 8 is what xcd_remap relies on),
    and xcd_remap over the virtual ids stays a bijection of the scale's tiles."""
    def table(grids):
        first = [0]
        for gx, gy, gz in grids:
            first.append(first[-1] + ((gx * gy * gz + 7) & ~7))
        return first

    def locate(first, grids, bid):
        s = 0
        while s < len(grids) - 1 and bid >= first[s + 1]:
            s += 1
        li = bid - first[s]
        gx, gy, gz = grids[s]
        if li >= gx * gy * gz:
            return None
        q = li // gx
        return s, li - q * gx, q - (q // gy) * gy, q // gy

    def xcd_remap(lin, total):
        q, r = total >> 3, total & 7
        xcd, k = lin & 7, lin >> 3
        return xcd * q + min(xcd, r) + k

    cdiv = lambda a, b: (a + b - 1) // b
    B = 16
    sizes = [(256, 832), (128, 416), (64, 208)]
    flat = lambda n: min(max(cdiv(n, 256), 1), 8192)
    families = {
        'flat (occlusion weights)': [(flat(B // 2 * h * w), 1, 1) for h, w in sizes],
        'partial sums (masked mean, consistency)': [(cdiv(h * w, 2048), B, 1) for h, w in sizes],
        'smoothness tiles': [(cdiv(w, 64), cdiv(h, 8), B) for h, w in sizes],
        'SSIM strips x chunks': [(cdiv(w, 124) * cdiv(h, 16 if h >= 128 else 8), B, 1) for h, w in sizes],
        'ragged': [(3, 5, 2), (1, 1, 1), (7, 1, 3), (2, 2, 2)],
    }
    for name, grids in families.items():
        first = table(grids)
        assert all(f % 8 == 0 for f in first), name
        seen = {}
        for bid in range(first[-1]):
            v = locate(first, grids, bid)
            if v is not None:
                assert v not in seen, (name, v)
                seen[v] = bid
                assert (bid - first[v[0]]) % 8 == bid % 8                          # the XCD of the virtual id is the real one
        want = {(s, x, y, z) for s, (gx, gy, gz) in enumerate(grids) for x in range(gx) for y in range(gy) for z in range(gz)}
        assert set(seen) == want, name
        for s, (gx, gy, gz) in enumerate(grids):
            total = gx * gy * gz
            assert sorted(xcd_remap(x + gx * (y + gy * z), total) for x in range(gx) for y in range(gy) for z in range(gz)) == list(range(total))


def test_torch_library_is_built_in_tree():
    """build() also produces libunflow_torch.so (TORCH_LIBRARY(unflow_hip)); it resolves libunflow_hip.so through $ORIGIN."""
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    from unopticalflow_amd import torch_ops
    assert os.path.exists(torch_ops.LIB_PATH)
    dyn = subprocess.run(['readelf', '-d', torch_ops.LIB_PATH], capture_output=True, text=True).stdout
    assert 'libunflow_hip.so' in dyn and '$ORIGIN' in dyn, dyn


def test_host_library_has_no_hip_dependency():
    """libunflow_host.so (PNG unfilter for the DataLoader workers) must load without the HIP runtime."""
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    from unopticalflow_amd import _lib
    lib = _lib.load_host()
    assert hasattr(lib, 'unflow_png_unfilter')
    needed = subprocess.run(['readelf', '-d', _lib.HOST_LIB_PATH], capture_output=True, text=True).stdout
    assert 'amdhip' not in needed and 'hsa-runtime' not in needed, needed
    import numpy as np
    rows = np.array([[0, 1, 2, 3], [2, 5, 5, 5]], np.uint8)          # filter 0 (none), filter 2 (up)
    assert lib.unflow_png_unfilter(ctypes.c_void_p(rows.ctypes.data), 2, 3, 1) == 0
    assert rows[1, 1:].tolist() == [6, 7, 8]


def _build_capi_bench(tmp_path):
    import subprocess
    import __graft_entry__ as ge
    ge.build()
    exe = str(tmp_path / 'capi_bench')
    r = subprocess.run([os.environ.get('HIPCC', '/opt/rocm/bin/hipcc'), '-O1', '--offload-arch=gfx950',
                        os.path.join(ROOT, 'tools', 'capi_bench.cpp'), '-I' + os.path.join(ROOT, 'include'),
                        '-L' + os.path.join(ROOT, 'unopticalflow_amd'), '-lunflow_hip',
                        '-Wl,-rpath,' + os.path.join(ROOT, 'unopticalflow_amd'), '-o', exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    assert os.path.exists(exe)
    return exe


def test_c_program_links_against_the_abi(tmp_path):
    """tools/capi_bench.cpp uses include/unflow_hip.h from plain C++ (no torch, no Python): it must compile and
    link against the in-tree library (it is RUN by test_c_program_runs_and_matches_the_oracle on the GPU box)."""
    _build_capi_bench(tmp_path)


def test_one_abi_version_number():
    """include/unflow_hip.h's UNFLOW_ABI_VERSION is what the library returns (csrc/photo.hip returns the macro), what the Python
    binding expects (_lib.ABI_VERSION is read from the header) and what the C program compares with."""
    from unopticalflow_amd import _lib
    assert _lib.ABI_VERSION == _lib.header_abi_version() == _lib.BINDING_ABI >= 10          # library == header == ctypes table
    assert 'return UNFLOW_ABI_VERSION;' in open(os.path.join(ROOT, 'unopticalflow_amd', 'csrc', 'photo.hip')).read()
    assert 'unflow_abi_version() != UNFLOW_ABI_VERSION' in open(os.path.join(ROOT, 'tools', 'capi_bench.cpp')).read()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    lib.unflow_abi_version.restype = ctypes.c_int
    assert lib.unflow_abi_version() == _lib.ABI_VERSION          # (a host function: callable without a GPU)


@pytest.mark.gpu
def test_c_program_runs_and_matches_the_oracle(tmp_path):
    """The Python-free user of the boundary ("standalone bench/rocprof harness without Python"), RUN as a fresh child
    process: cost volume forward + backward through the C ABI on its deterministic input, compared with the CPU oracle's
    corr_naive (pwc_tf.py:97-106) and its autograd."""
    import re
    import subprocess
    import numpy as np
    from oracle import ref_cpu as R
    exe = _build_capi_bench(tmp_path)
    B, C, H, W, d = 2, 12, 24, 68, 4
    mask_file = str(tmp_path / 'mask.bin')
    r = subprocess.run([exe] + [str(v) for v in (B, C, H, W, d, 2)] + [mask_file], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-2000:])
    got = {m.group(1): [float(x) for x in m.group(2).split()] for m in re.finditer(r'^(\w+) = (.*)$', r.stdout, re.M)}
    D = 2 * d + 1
    nf, nc = B * C * H * W, B * D * D * H * W
    i = np.arange(max(nf, nc) + 16, dtype=np.uint64)
    v = ((i * np.uint64(2654435761)) % np.uint64(2001)).astype(np.float32) / np.float32(1000.0) - np.float32(1.0)
    f1 = torch.from_numpy(v[:nf].reshape(B, C, H, W).copy()).requires_grad_()
    f2 = torch.from_numpy(v[7:7 + nf].reshape(B, C, H, W).copy()).requires_grad_()
    g = torch.from_numpy(v[3:3 + nc].reshape(B, D * D, H, W).copy())
    cv = R.corr_naive(f1, f2, d)
    cv.backward(g)
    centre = cv[0, d * D + d, H // 2, :4].detach().numpy()
    np.testing.assert_allclose(got['cv_centre'], centre, rtol=1e-5, atol=1e-6)
    for name, t in (('sum_abs_cv', cv.detach()), ('sum_abs_gf1', f1.grad), ('sum_abs_gf2', f2.grad)):
        np.testing.assert_allclose(got[name][0], t.double().abs().sum().item(), rtol=1e-5, err_msg=name)
    assert 'GB/s algorithmic' in r.stdout
    # the integer half of the parity bar through the same Python-free boundary: the binary validity mask of the masked image warp
    # (net_utils.py:47-52), byte for byte against the oracle, and the warped image
    img = torch.from_numpy(((v[11:11 + B * 3 * H * W] + np.float32(1.0)) / np.float32(2.0)).reshape(B, 3, H, W).copy())
    fl = torch.from_numpy((np.float32(6.0) * v[5:5 + B * 2 * H * W]).reshape(B, 2, H, W).copy())
    m_ref = R.warp_mask(img.shape, fl, False).numpy().astype(np.uint8).reshape(-1)
    m_got = np.fromfile(mask_file, dtype=np.uint8)
    assert got['mask_not_binary'][0] == 0 and 0 < got['mask_ones'][0] < m_ref.size      # (a flow of +-6 px leaves both kinds of pixels)
    assert np.array_equal(m_got, m_ref) and got['mask_ones'][0] == int(m_ref.sum())
    np.testing.assert_allclose(got['sum_abs_warped'][0], R.warp_flow(img, fl, True, False).double().abs().sum().item(), rtol=1e-5)


def test_miopen_tuning_paths(monkeypatch):
    """tuning.enable_miopen_tuning: every process gets a private copy of the shipped find-db (the package directory is
    never written); find mode only when the shipped files are named for this device AND this MIOpen build; a
    user-provided MIOPEN_USER_DB_PATH is left alone."""
    import torch
    from unopticalflow_amd import tuning
    saved = torch.backends.cudnn.benchmark
    try:
        shipped = sorted(os.listdir(tuning.DB_DIR))
        assert any(n.endswith('.ufdb.txt') for n in shipped) and any(n.endswith('.udb.txt') for n in shipped)
        monkeypatch.delenv('MIOPEN_USER_DB_PATH', raising=False)
        monkeypatch.delenv('UNFLOW_MIOPEN_FORCE_FIND', raising=False)
        private = tuning.enable_miopen_tuning()
        assert private != tuning.DB_DIR and os.path.dirname(private) != os.path.dirname(tuning.DB_DIR)
        assert sorted(os.listdir(private)) == shipped
        assert not torch.backends.cudnn.benchmark          # no GPU here: the device key cannot match
        monkeypatch.delenv('MIOPEN_USER_DB_PATH', raising=False)
        other = tuning.enable_miopen_tuning()
        assert other != private                             # one copy per call / process, never shared
        # the name check: device key + MIOpen build tag must both match a shipped file
        dev_key, tag = shipped[0].split('.HIP.')[0], 'HIP.' + shipped[0].split('.HIP.')[1].rsplit('.', 2)[0]
        assert tuning.shipped_db_matches(dev_key, tag)
        assert not tuning.shipped_db_matches(dev_key, tag + 'x')             # another MIOpen build
        assert not tuning.shipped_db_matches(dev_key, 'HIP.3_5_1_5b515cf1bc')
        assert not tuning.shipped_db_matches('gfx95080', tag)                # another partition mode
        assert not tuning.shipped_db_matches(None, tag) and not tuning.shipped_db_matches(dev_key, None)
        t = tuning.loaded_miopen_db_tag()                   # torch's own libMIOpen is mapped in this process
        assert t is None or t.startswith('HIP.')
        monkeypatch.setenv('MIOPEN_USER_DB_PATH', '/some/where')
        assert tuning.enable_miopen_tuning(benchmark=False) == '/some/where' and not torch.backends.cudnn.benchmark
    finally:
        torch.backends.cudnn.benchmark = saved


def test_channels_last_stride_helpers():
    """Host logic of the channels_last epilogue (no GPU): which gradients can be read in place as pixel-strided NHWC data (a
    dense channels_last tensor, or a 4-aligned channel slice of one -- what autograd hands out for a torch.cat operand) and
    which have to be re-laid out first."""
    from unopticalflow_amd import ops
    CL = torch.channels_last
    x = torch.zeros(2, 8, 5, 6).contiguous(memory_format=CL)
    assert ops._is_nhwc(x) and not ops._is_nhwc(torch.zeros(2, 8, 5, 6)) and not ops._is_nhwc(torch.zeros(2, 1, 5, 6).contiguous(memory_format=CL))
    g, ps = ops._pixel_strided(x, (2, 8, 5, 6))
    assert g is x and ps == 8
    wide = torch.zeros(2, 24, 5, 6).contiguous(memory_format=CL)
    g, ps = ops._pixel_strided(wide[:, 8:16], (2, 8, 5, 6))            # channel slice at a 16-byte aligned offset: read in place
    assert g.data_ptr() == wide[:, 8:16].data_ptr() and ps == 24
    g, ps = ops._pixel_strided(wide[:, 3:11], (2, 8, 5, 6))            # misaligned slice: copied
    assert ps == 8 and g.is_contiguous(memory_format=CL) and g.data_ptr() != wide[:, 3:11].data_ptr()
    g, ps = ops._pixel_strided(torch.zeros(2, 8, 5, 6), (2, 8, 5, 6))   # an NCHW gradient from outside the island: re-laid out
    assert ps == 8 and g.is_contiguous(memory_format=CL)
    g, st = ops._sample_strided(torch.zeros(2, 24, 5, 6)[:, 8:16], (2, 8, 5, 6))   # the NCHW twin: sample-strided slice
    assert st == 24 * 30 or g.is_contiguous()
