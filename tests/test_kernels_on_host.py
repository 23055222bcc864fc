"""The flat `_ms` kernels of csrc/multiscale.h EXECUTED on the build host (tests/host_check/ms_flat_check.cpp compiles the shipped
definitions -- csrc/ms_flat_photo.h, csrc/ms_flat_warp.h, csrc/multiscale.h, csrc/bodies/*.inc -- with g++ and runs every workgroup and
lane): one launch over three scales leaves, bit for bit, what three single-scale launches leave (checked inside the program), and what it
leaves is what the oracle computes (checked here).  Five of the twelve `_ms` kernels can run this way (no LDS, barrier or wave shuffle in
their bodies); all twelve share the prologue and the workgroup table this executes."""
import os
import subprocess

import numpy as np
import torch

from oracle import ref_cpu as R

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_san_builds = {}
# UNFLOW_HOST_CHECK_SANITIZE: unset = the check programs run once, plain -O2 (the default CPU tier: the AddressSanitizer + UBSan compiles of the kernel
# files alone were ~70 s of it); '1' = each program once more under AddressSanitizer + UBSan (run after every change of a kernel file, last: see
# profiles/README.md); 'all' = also the long ones (warp at ~100 s, -O2 against instrumented -O1 bytes of the cost-volume program)
_LEVEL = os.environ.get('UNFLOW_HOST_CHECK_SANITIZE', '')


def _sanitize(always):
    return _LEVEL == 'all' or (always and _LEVEL == '1')


def _sources_key():
    """sha256 over everything a check program is made of: csrc/ (kernel files, headers, bodies), tests/host_check/ sources and tools/proto headers."""
    import hashlib
    h = hashlib.sha256()
    for d in (os.path.join(ROOT, 'unopticalflow_amd', 'csrc'), os.path.join(ROOT, 'unopticalflow_amd', 'csrc', 'bodies'),
              os.path.join(ROOT, 'tests', 'host_check'), os.path.join(ROOT, 'tools', 'proto'), os.path.join(ROOT, 'include')):
        for f in sorted(os.listdir(d)):
            if f.endswith(('.h', '.hip', '.inc', '.cpp')):
                h.update(f.encode()); h.update(open(os.path.join(d, f), 'rb').read())
    return h.hexdigest()[:20]


def _build_cached(build_cmd, builder=None):
    """Run a check program's compile command -- or copy the program from tests/host_check/_build/programs/ when this very command was already run on
    these very sources (a second pass over an unchanged tree then skips ~25 s of compiling corr.hip / warp.hip for the host).  The command's output
    path is its `-o` argument; the key is the command with the temporary directory taken out + _sources_key().  -> CompletedProcess-like (returncode, stderr)."""
    import hashlib
    import shutil
    import types
    exe = build_cmd[build_cmd.index('-o') + 1]
    tmp = os.path.dirname(exe)
    norm = ' '.join(a.replace(tmp, '@') for a in build_cmd)
    key = hashlib.sha256((norm + _sources_key()).encode()).hexdigest()[:24]
    store = os.path.join(ROOT, 'tests', 'host_check', '_build', 'programs')
    hit = os.path.join(store, key)
    if os.path.exists(hit):
        shutil.copy2(hit, exe)
        return types.SimpleNamespace(returncode=0, stderr='')
    r = builder() if builder is not None else subprocess.run(build_cmd, capture_output=True, text=True)      # (builder: several compilers side by side + a link, described by build_cmd for the key)
    if r.returncode == 0:
        os.makedirs(store, exist_ok=True)
        old = sorted((os.path.getmtime(os.path.join(store, f)), f) for f in os.listdir(store))
        for _, f in old[:-40]:                                   # (keep the store bounded)
            os.remove(os.path.join(store, f))
        shutil.copy2(exe, hit + '.tmp%d' % os.getpid())
        os.replace(hit + '.tmp%d' % os.getpid(), hit)
    return r


def _sanitized_build_started(build_cmd, tmp_path, name, always=True):
    """Start the sanitized build of a check program NOW, next to the plain one's build and run (_sanitized() then waits for it)."""
    if not _sanitize(always):
        return
    exe = str(tmp_path / (name + '_asan'))
    cmd = [a for a in build_cmd if a not in ('-O2',)]
    cmd = cmd[:1] + ['-O1', '-g', '-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-fno-sanitize-recover=undefined'] + [a for a in cmd[1:] if a != '-O1']
    cmd[cmd.index('-o') + 1] = exe
    _san_builds[exe] = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)


def _sanitized(build_cmd, run_args, tmp_path, name, always=True):
    """The same program once more under AddressSanitizer + UndefinedBehaviorSanitizer (the CPU build is where sanitizers run: no GPU ASan on
    this pool): every global-memory access of the executed kernels lands inside the buffers it was given (they are exactly-sized heap
    blocks with red zones around them), every LDS access inside its array, no signed overflow / misaligned access / bad shift in the
    index arithmetic.  `always` = False: only with UNFLOW_HOST_CHECK_SANITIZE=all (the long ones); nothing without UNFLOW_HOST_CHECK_SANITIZE."""
    if not _sanitize(always):
        return
    exe = str(tmp_path / (name + '_asan'))
    if exe not in _san_builds:
        _sanitized_build_started(build_cmd, tmp_path, name)
    pr = _san_builds.pop(exe)
    err = pr.communicate()[1]
    assert pr.returncode == 0, err[-3000:]
    env = dict(os.environ, ASAN_OPTIONS='detect_stack_use_after_return=0:detect_leaks=0')      # (the lanes' stacks are heap blocks switched by hand)
    r = subprocess.run([exe] + list(run_args), capture_output=True, text=True, timeout=3000, env=env)
    assert r.returncode == 0 and 'ERROR' not in r.stderr and 'runtime error' not in r.stderr, (r.stdout[-1500:], r.stderr[-3000:])


def _v(n, seed, scale=1.0, shift=0.0):
    i = np.arange(n, dtype=np.uint64) + np.uint64(seed)
    a = ((i * np.uint64(2654435761)) % np.uint64(2001)).astype(np.float32) / np.float32(1000.0) - np.float32(1.0)
    return torch.from_numpy(a * np.float32(scale) + np.float32(shift))


def test_flat_ms_kernels_run_on_the_host_and_match_the_oracle(tmp_path):
    exe, out = str(tmp_path / 'ms_flat_check'), str(tmp_path / 'out.bin')
    build = ['g++', '-O1', '-std=c++17', '-ffp-contract=off', '-DUNFLOW_HOST_CHECK', '-I', os.path.join(ROOT, 'unopticalflow_amd', 'csrc'),
                        os.path.join(ROOT, 'tests', 'host_check', 'ms_flat_check.cpp'), '-o', exe]
    _sanitized_build_started(build, tmp_path, 'ms_flat_check')
    r = _build_cached(build)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and 'OK: 0 mismatches' in r.stdout, r.stdout[-2000:]          # one launch over the scales == three launches, bit for bit
    _sanitized(build, [str(tmp_path / 'san.bin')], tmp_path, 'ms_flat_check')
    raw = open(out, 'rb').read()
    pos = [0]

    def take(shape, dtype=np.float32):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = np.frombuffer(raw[pos[0]:pos[0] + n], dtype=dtype).reshape(shape)
        pos[0] += n
        return torch.from_numpy(a.copy())

    n, B = 3, 2
    Hs, Ws = (12, 6, 5), (70, 36, 17)
    S = []
    for s in range(n):
        H, W = Hs[s], Ws[s]
        hw = H * W
        img = _v(B * 3 * hw, 11 + s, 0.5, 0.5).view(B, 3, H, W)
        warped = _v(2 * B * 3 * hw, 101 + s, 0.5, 0.5).view(2 * B, 3, H, W).clone()
        flat = warped.view(-1)
        for c in range(3):
            flat[c * hw:c * hw + hw // 3] = 0.0
        S.append(dict(H=H, W=W, hw=hw, img=img, warped=warped, flow=_v(B * 2 * hw, 201 + s, 4.0).view(B, 2, H, W),
                      gdiff=_v(2 * B * hw, 301 + s).view(2 * B, 1, H, W), sums=_v(2 * B * 2, 401 + s, 0.25, 0.75 * hw).view(2 * B, 2),
                      gloss=_v(2 * B, 501 + s), ff=_v(B * 2 * hw, 601 + s, 3.0).view(B, 2, H, W), fb=_v(B * 2 * hw, 701 + s, 3.0).view(B, 2, H, W),
                      w=_v(B * hw, 801 + s, 1.0, 1.0).view(B, 1, H, W)))
    close = lambda a, b, what, tol=2e-6: np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=1e-5, atol=tol * max(float(b.abs().max()), 1e-30), err_msg=what)
    for ac in (False, True):                                               # the masked image warp and its binary mask, both grid_sample conventions
        for s, q in enumerate(S):
            wout, mask = take((B, 3, q['H'], q['W'])), take((B, 1, q['H'], q['W']), np.uint8)
            assert torch.equal(mask, R.warp_mask(q['img'].shape, q['flow'], ac)), 'mask, scale %d' % s    # bit-exact target #1 (net_utils.py:47-51)
            assert 0 < int(mask.sum()) < mask.numel()
            close(wout, R.warp_flow(q['img'], q['flow'], True, ac), 'warped image, scale %d' % s)
    for s, q in enumerate(S):
        H, W, hw = q['H'], q['W'], q['hw']
        diff, wgt = take((2 * B, 1, H, W)), take((2 * B, 1, H, W))
        gfrom, gmm, gflow = take((2 * B, 3, H, W)), take((2 * B, 1, H, W)), take((B, 2, H, W))
        d_l, d_r, w_b, w_f, _, _ = R.diff_weight(q['img'], q['warped'][:B], q['warped'][B:])
        close(diff, torch.cat((d_l, d_r)), 'diff, scale %d' % s)
        close(wgt, torch.cat((w_b, w_f)), 'occlusion weight, scale %d' % s, tol=5e-6)
        assert float(wgt[0, 0].view(-1)[:hw // 3].abs().max()) == 0.0                               # the all-zero region is invalid: weight 0
        f = q['warped'].clone().requires_grad_()
        torch.abs(q['img'].repeat(2, 1, 1, 1) - f).mean(1, True).backward(q['gdiff'])
        close(gfrom, f.grad, '|.| backward, scale %d' % s)
        k = q['gloss'] / hw / (q['sums'][:, 1] / hw + 1e-12)
        close(gmm, k.view(-1, 1, 1, 1) * wgt, 'masked-mean backward, scale %d' % s)
        a = q['ff'].clone().requires_grad_()
        occ = 1 - q['w']
        num = (torch.abs(R.flow_normalization(a) + R.flow_normalization(q['fb'])) * occ).sum((1, 2, 3)) / (2.0 * hw)
        (num / (q['sums'][:B, 1] / hw + 1e-12)).backward(q['gloss'][:B])
        close(gflow, a.grad, 'consistency backward, scale %d' % s, tol=1e-5)
        vb, vf = take((B, 1, H, W), np.uint8), take((B, 1, H, W), np.uint8)
        _, _, _, _, rb, rf = R.diff_weight(q['img'], q['warped'][:B], q['warped'][B:])
        assert torch.equal(vb, rb.to(torch.uint8)) and torch.equal(vf, rf.to(torch.uint8)), 'validity masks, scale %d' % s   # bit-exact target #2 (:111-112)
        assert int(vb.sum()) < vb.numel()                                                  # (the all-zero region is in the bwd direction)
    assert pos[0] == len(raw)


def test_input_stage_kernel_runs_on_the_host_bit_exact(tmp_path):
    """csrc/prepare.hip itself, compiled with g++ and executed lane by lane (tests/host_check/prepare_check.cpp), against
    oracle/prepare_cpu.py: the byte / integer arithmetic of the input stage (cv2's fixed-point 8-bit resize, flip, / 255, BGR planes) is
    bit-exact on the build host too -- KITTI's native sizes, up- and down-scaling, odd sizes, leftover rows, flips, RGB-decoded sources."""
    import struct
    from oracle.prepare_cpu import prepare_triplet
    exe = str(tmp_path / 'prepare_check')
    build = ['g++', '-O1', '-std=c++20', '-ffp-contract=off', '-DUNFLOW_HOST_CHECK', '-I', os.path.join(ROOT, 'tests', 'host_check'),
             '-I', os.path.join(ROOT, 'unopticalflow_amd', 'csrc'), os.path.join(ROOT, 'tests', 'host_check', 'prepare_check.cpp'), '-o', exe]
    r = _build_cached(build)
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(11)
    for (H, W), swap in (((64, 128), 0), ((32, 52), 1)):
        sizes = [(375, 1242), (370, 1226), (H, W), (2 * H, 2 * W), (40, 50), (H + 1, 3 * W + 7)]
        images = [rng.integers(0, 256, (3 * h + (i % 3), w, 3), dtype=np.uint8) for i, (h, w) in enumerate(sizes)]
        flips = [bool(i & 1) for i in range(len(images))]
        fin, fout = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
        with open(fin, 'wb') as f:
            f.write(struct.pack('4i', len(images), H, W, swap))
            for im, fl in zip(images, flips):
                f.write(struct.pack('3i', im.shape[0], im.shape[1], int(fl)))
            for im in images:
                f.write(np.ascontiguousarray(im).tobytes())
        r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout, r.stderr)
        _sanitized(build, [fin, str(tmp_path / 'san.bin')], tmp_path, 'prepare_check', always=False)
        out = np.fromfile(fout, dtype=np.float32).reshape(len(images), 3, 3 * H, W)
        for i, im in enumerate(images):
            src = im[:, :, ::-1] if swap else im                              # swap_rb: an RGB-decoded source lands in cv2's BGR planes
            np.testing.assert_array_equal(out[i], prepare_triplet(src, (H, W), flips[i]), err_msg='image %d, %dx%d' % (i, H, W))
    # ... and against the REFERENCE's own input pipeline (tests/golden/g7_prepare.npz: KITTI_Prepared.preprocess_img + the tail of __getitem__,
    # kitti_prepared.py:63-99,146-154, at the image's native size): the kernel source leaves its bytes, flipped and not
    g = dict(np.load(os.path.join(ROOT, 'tests', 'golden', 'g7_prepare.npz')))
    H, W = (int(v) for v in g['img_hw'])
    with open(fin, 'wb') as f:
        f.write(struct.pack('4i', 2, H, W, 0))
        for fl in (0, 1):
            f.write(struct.pack('3i', g['img'].shape[0], g['img'].shape[1], fl))
        for _ in (0, 1):
            f.write(np.ascontiguousarray(g['img']).tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout, r.stderr)
    out = np.fromfile(fout, dtype=np.float32).reshape(2, 3, 3 * H, W)
    for fl in (0, 1):
        np.testing.assert_array_equal(out[fl], g['out_flip%d' % fl], err_msg='reference fixture, flip %d' % fl)


def test_loss_kernels_run_on_the_host_and_match_the_oracle(tmp_path):
    """csrc/photo.hip ITSELF -- occlusion weights, masked mean, second-order smoothness (the LDS-tile kernels), consistency, the batched
    second stage, forward and backward, single-scale and `_ms`, kernels AND C entries -- compiled with g++ and executed with lanes as fibers
    (tests/host_check/hip_on_host.h: real barriers, the butterfly reductions of common.h as written).  Inside the program: one launch per
    loss over three scales leaves the bits of the scale-by-scale entries everywhere, partial sums included.  Here: what it leaves is what
    the oracle computes -- three losses per scale and every gradient."""
    exe, out = str(tmp_path / 'photo_check'), str(tmp_path / 'out.bin')
    build = ['g++', '-O1', '-std=c++20', '-ffp-contract=off', '-DUNFLOW_HOST_CHECK', '-I', os.path.join(ROOT, 'tests', 'host_check'),
                        '-I', os.path.join(ROOT, 'unopticalflow_amd', 'csrc'), os.path.join(ROOT, 'tests', 'host_check', 'photo_check.cpp'), '-o', exe]
    _sanitized_build_started(build, tmp_path, 'photo_check')
    r = _build_cached(build)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'OK: 0 mismatches' in r.stdout, r.stdout[-2000:]
    raw = np.fromfile(out, dtype=np.float32)
    _sanitized(build, [str(tmp_path / 'san.bin')], tmp_path, 'photo_check')
    pos = [0]

    def take(*shape):
        n = int(np.prod(shape))
        a = torch.from_numpy(raw[pos[0]:pos[0] + n].reshape(shape).copy())
        pos[0] += n
        return a

    B = 2
    for s, (H, W) in enumerate(((20, 136), (10, 68), (5, 34))):
        hw = H * W
        img = _v(B * 3 * hw, 11 + s, 0.5, 0.5).view(B, 3, H, W)
        warped = _v(2 * B * 3 * hw, 101 + s, 0.5, 0.5).view(2 * B, 3, H, W).clone()
        for c in range(3):
            warped.view(-1)[c * hw:c * hw + hw // 3] = 0.0
        flows = _v(2 * B * 2 * hw, 201 + s, 3.0).view(2 * B, 2, H, W)
        gl_pix, gl_sm, gl_co = _v(2 * B, 301 + s), _v(2 * B, 311 + s), _v(B, 321 + s)
        l_pix, l_sm, l_co = take(2 * B), take(2 * B), take(B)
        gfrom, gflow_sm, gflow_co = take(2 * B, 3, H, W), take(2 * B, 2, H, W), take(B, 2, H, W)
        wp, fl = warped.clone().requires_grad_(), flows.clone().requires_grad_()
        d_l, d_r, w_b, w_f, _, _ = R.diff_weight(img, wp[:B], wp[B:])
        r_pix = torch.cat((R.masked_l1(d_l, w_b), R.masked_l1(d_r, w_f)))
        r_sm = R.grad2_error(fl / 20.0, img.repeat(2, 1, 1, 1))
        close = lambda a, b, what, rtol=2e-5, tol=2e-6: np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=rtol, atol=tol * max(float(b.abs().max()), 1e-30),
                                                                                err_msg='%s, scale %d' % (what, s))
        close(l_pix, r_pix, 'masked mean'); close(l_sm, r_sm, 'smoothness')
        (r_pix * gl_pix).sum().backward()
        close(gfrom, wp.grad, '|.| + masked-mean backward', rtol=1e-4)
        (r_sm * gl_sm).sum().backward()
        close(gflow_sm, fl.grad, 'smoothness backward', rtol=1e-4, tol=5e-6)
        ff = flows[B:].clone().requires_grad_()
        r_co = R.consis_loss(ff, flows[:B], w_f.detach())
        close(l_co, r_co, 'consistency')
        (r_co * gl_co).sum().backward()
        close(gflow_co, ff.grad, 'consistency backward', rtol=1e-4, tol=1e-5)
    assert pos[0] == raw.size


def test_ssim_kernels_run_on_the_host_on_flat_patches(tmp_path):
    """csrc/ssim.hip ITSELF (the sum-space column-pair kernels of the train step, their `_ms` forms, the general kernels of odd widths; kernels
    and C entries) compiled for the host with the ROCm clang++ and executed with lanes as fibers, on the inputs VERDICT r4 asked for:
    saturated flat patches (1.0 against 1.0, 1 - 1/255, 0.95), dark flat patches, a step edge.  Inside the program the `_ms` launch equals
    three single-scale launches bit for bit; here loss and gradient are held to the oracle (pytorch_ssim/ssim.py:4-20,
    model_flow_paper.py:137-148) at the GPU tests' bars: loss 1e-4 rel, gradient 1e-4 + 2e-5 of the largest."""
    clang = '/opt/rocm/lib/llvm/bin/clang++'
    if not os.path.exists(clang):
        import pytest
        pytest.skip('the SSIM kernels use clang vector extensions: no ROCm clang++ here')
    exe, out = str(tmp_path / 'ssim_check'), str(tmp_path / 'out.bin')
    build = [clang, '-O1', '-std=c++20', '-ffp-contract=off', '-DUNFLOW_HOST_CHECK', '-Wno-unknown-attributes', '-Wno-unknown-pragmas', '-Wno-pass-failed',
                        '-I', os.path.join(ROOT, 'tests', 'host_check'), '-I', os.path.join(ROOT, 'unopticalflow_amd', 'csrc'),
                        os.path.join(ROOT, 'tests', 'host_check', 'ssim_check.cpp'), '-o', exe]
    _sanitized_build_started(build, tmp_path, 'ssim_check')
    r = _build_cached(build)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run([exe, out], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and 'OK: 0 mismatches' in r.stdout, r.stdout[-2000:]
    _sanitized(build, [str(tmp_path / 'san.bin')], tmp_path, 'ssim_check')
    raw = np.fromfile(out, dtype=np.float32)
    pos = [0]

    def take(*shape):
        n = int(np.prod(shape))
        a = torch.from_numpy(raw[pos[0]:pos[0] + n].reshape(shape).copy())
        pos[0] += n
        return a

    B, B2 = 1, 2
    for H, W in ((128, 64), (64, 32), (32, 16), (33, 57)):
        img, warped, w, gloss = take(B, 3, H, W), take(B2, 3, H, W), take(B2, 1, H, W), take(B2)
        loss, g = take(B2), take(B2, 3, H, W)
        assert float(img[0, :, 0, 0].min()) == 1.0 and float(warped[0, 0, 0, W // 4 if W // 4 > 2 else 2]) == np.float32(1.0 - 1.0 / 255.0)   # (the patches are there)
        y = warped.clone().requires_grad_()
        ref = R.ssim_loss(img.repeat(B2 // B, 1, 1, 1), y, w)
        np.testing.assert_allclose(loss.numpy(), ref.detach().numpy(), rtol=1e-4, err_msg='SSIM loss %dx%d' % (H, W))
        (ref * gloss).sum().backward()
        np.testing.assert_allclose(g.numpy(), y.grad.numpy(), rtol=1e-4, atol=2e-5 * float(y.grad.abs().max()), err_msg='SSIM gradient %dx%d' % (H, W))
    assert pos[0] == raw.size


def _structured_flow(B, h, w, kind, seed):
    """As tests/test_hip_ops.py: flows that take every branch of the LDS-tile warps ('smooth': the tile's source window fits LDS; 'mixed': a
    noisy band, some tiles fall back to per-tap gathers; 'outside': empty windows; 'edge': windows clipped by the border; 'noise': no tile fits)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.meshgrid(np.arange(h, dtype=np.float32), np.arange(w, dtype=np.float32), indexing='ij')
    fl = np.stack((2.3 + 1.5 * np.sin(xx / 37.0) * np.cos(yy / 23.0), -1.1 + 0.8 * np.cos(xx / 29.0 + yy / 41.0)), 0)[None].repeat(B, 0).astype(np.float32)
    fl += rng.standard_normal((B, 2, 1, 1)).astype(np.float32)
    if kind == 'mixed':
        fl[:, :, h // 3: h // 3 + max(2, h // 6)] += (rng.standard_normal((B, 2, max(2, h // 6), w)) * 9).astype(np.float32)
    elif kind == 'outside':
        fl[:, 0] += 3.0 * w
    elif kind == 'edge':
        fl[:, 0] += w / 2.0 - 3.0
        fl[:, 1] -= h / 2.0
    elif kind == 'noise':
        fl = (rng.standard_normal((B, 2, h, w)) * 7).astype(np.float32)
    return torch.from_numpy(fl)


def test_warp_kernels_run_on_the_host_and_match_the_oracle(tmp_path):
    """csrc/warp.hip ITSELF -- LDS-tile forward, tile / cell scatter backward, the one-pass gather backward with the table its forward leaves
    (what ops.warp_flow picks BY ITSELF at level 2 of the 832x256 step: VERDICT r4's parity hole 3b, here at that launch's shape
    with half the batch, [8,32,64,208]), the row-segment image warps with their binary masks, and the `_ms` image warps -- compiled with g++ and executed with
    lanes as fibers, against the oracle's grid_sample chain (net_utils.py:16-54): forward 1e-5, gradients 1e-4 + 2e-5 of the largest, masks
    bit for bit, for the five flow kinds of the GPU tests."""
    import struct
    exe = str(tmp_path / 'warp_check')
    build = ['g++', '-O2', '-std=c++20', '-ffp-contract=off', '-DUNFLOW_HOST_CHECK', '-I', os.path.join(ROOT, 'tests', 'host_check'),
                        '-I', os.path.join(ROOT, 'unopticalflow_amd', 'csrc'), os.path.join(ROOT, 'tests', 'host_check', 'warp_check.cpp'), '-o', exe]
    r = _build_cached(build)
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(5)
    cases = []                                                              # (B, C, H, W, masked, ac, flow kind)
    cases.append((8, 16, 64, 208, 0, 0, 'smooth'))                         # level 2 of the step at B = 4 pairs (half its channels): the one-pass backward is picked (supported == 2)
    for kind in ('mixed', 'outside', 'edge', 'noise'):
        cases.append((2, 32 if kind == 'mixed' else 16, 64, 208, 0, 0, kind))
    cases += [(3, 32, 32, 104, 0, 0, 'smooth'), (2, 64, 32, 104, 0, 1, 'mixed'), (2, 96, 16, 52, 0, 0, 'smooth'), (2, 128, 8, 26, 0, 0, 'edge'),
              (2, 9, 17, 130, 0, 1, 'noise'), (2, 20, 40, 72, 0, 0, 'mixed')]
    cases += [(3, 3, 40, 100, 1, 0, 'mixed'), (3, 3, 20, 50, 1, 0, 'edge'), (3, 3, 10, 25, 1, 0, 'noise')]      # an image pyramid: masked, no source gradient
    data = []
    fin, fout = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(fin, 'wb') as f:
        f.write(struct.pack('i', len(cases)))
        for k, (B, C, H, W, masked, ac, kind) in enumerate(cases):
            src = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32)) if not masked else torch.from_numpy(rng.random((B, C, H, W), dtype=np.float32))
            flow = _structured_flow(B, H, W, kind, seed=100 + k)
            gout = torch.from_numpy(rng.standard_normal((B, C, H, W)).astype(np.float32))
            data.append((src, flow, gout))
            f.write(struct.pack('6i', B, C, H, W, masked, ac))
            for t in (src, flow, gout):
                f.write(t.numpy().tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and 'OK: 0 mismatches' in r.stdout, (r.stdout[-2000:], r.stderr[-500:])      # (the `_ms` image warps == the per-scale ones, bit for bit)
    _sanitized(build, [fin, str(tmp_path / 'san.bin')], tmp_path, 'warp_check', always=False)      # (~100 s: UNFLOW_HOST_CHECK_SANITIZE=all)
    raw = open(fout, 'rb').read()
    pos = [0]

    def take(shape, dtype=np.float32):
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        a = np.frombuffer(raw[pos[0]:pos[0] + n], dtype=dtype).reshape(shape)
        pos[0] += n
        return torch.from_numpy(a.copy())

    def close(a, b, what, rtol, atol):
        np.testing.assert_allclose(a.numpy(), b.detach().numpy(), rtol=rtol, atol=atol, err_msg=what)

    picked = 0
    for k, ((B, C, H, W, masked, ac, kind), (src, flow, gout)) in enumerate(zip(cases, data)):
        what = 'case %d %s %s' % (k, (B, C, H, W), kind)
        x, fl = src.clone().requires_grad_(not masked), flow.clone().requires_grad_()
        ref = R.warp_flow(x, fl, bool(masked), bool(ac))
        ref.backward(gout)
        gf_tol = 2e-5 * max(float(fl.grad.abs().max()), 1e-6)
        close(take((B, C, H, W)), ref, what + ' forward', 1e-5, 1e-6)
        if masked:
            assert torch.equal(take((B, 1, H, W), np.uint8), R.warp_mask(src.shape, flow, bool(ac))), what + ' mask'
            close(take((B, 2, H, W)), fl.grad, what + ' flow gradient (masked)', 1e-4, gf_tol)
            continue
        close(take((B, C, H, W)), x.grad, what + ' source gradient (scatter)', 1e-4, 2e-5 * float(x.grad.abs().max()) + 1e-7)
        close(take((B, 2, H, W)), fl.grad, what + ' flow gradient', 1e-4, gf_tol)
        fused = int(take((1,), np.int32)[0])
        if fused:
            picked += fused == 2
            close(take((B, C, H, W)), ref, what + ' forward (table-leaving)', 1e-5, 1e-6)
            close(take((B, C, H, W)), x.grad, what + ' source gradient (one pass)', 1e-4, 2e-5 * float(x.grad.abs().max()) + 1e-7)
            close(take((B, 2, H, W)), fl.grad, what + ' flow gradient (one pass)', 1e-4, gf_tol)
    assert pos[0] == len(raw) and picked >= 1                               # (the level-2 launch took the one-pass form by itself)


def test_matrix_core_backward_runs_on_the_host(tmp_path):
    """The cost-volume backward on the matrix cores compiled for the host (ROCm clang++) and executed with lanes as fibers: the matrix
    instruction as a function that gathers the wave's A / B fragments by the CDNA4 lane layouts, range-checked buffer accesses, LDS tables.
    csrc/corr_mfma.h -- the shipped kernel (on request: ops.corr(..., backward='mfma')); with UNFLOW_HOST_CHECK_PROTO=1 also the never-run pixel-pair
    prototype tools/proto/corr_mfma2.h -- against the oracle's autograd of corr_naive (pwc_tf.py:97-106) at the GPU test's bar (rtol 1e-4 + 1e-5 of the largest
    gradient): both radii, ragged last segments, a partial channel group, chunks that do not divide the rows, one row chunk and several."""
    import struct
    clang = '/opt/rocm/lib/llvm/bin/clang++'
    if not os.path.exists(clang):
        import pytest
        pytest.skip('needs clang (vector extensions, __bf16): no ROCm clang++ here')
    exe = str(tmp_path / 'mfma_check')
    build = [clang, '-O2', '-std=c++20', '-ffp-contract=off', '-DUNFLOW_HOST_CHECK', '-Wno-unknown-attributes', '-Wno-unknown-pragmas', '-Wno-pass-failed',
                        '-I', os.path.join(ROOT, 'tests', 'host_check'), '-I', os.path.join(ROOT, 'unopticalflow_amd', 'csrc'),
                        os.path.join(ROOT, 'tests', 'host_check', 'mfma_check.cpp'), '-o', exe]
    proto = os.environ.get('UNFLOW_HOST_CHECK_PROTO') == '1'
    if proto:
        build[-3:-3] = ['-DWITH_PROTO', '-I', os.path.join(ROOT, 'tools', 'proto')]
    _sanitized_build_started(build, tmp_path, 'mfma_check')
    r = _build_cached(build)
    assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(3)
    shapes = [(4, 2, 32, 12, 48, 8), (4, 1, 48, 21, 100, 16), (4, 2, 16, 9, 36, 4), (8, 1, 32, 20, 48, 16), (8, 1, 16, 37, 44, 32)]       # R, B, C, H, W, rows per wave
    cases = [(s, which) for which in ((0, 1) if proto else (0,)) for s in shapes]
    data = []
    fin, fout = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(fin, 'wb') as f:
        f.write(struct.pack('i', len(cases)))
        for (R_, B, C, H, W, rows), which in cases:
            DD = 2 * R_ + 1
            f1, f2 = (torch.from_numpy(rng.uniform(-1, 1, (B, C, H, W)).astype(np.float32)) for _ in range(2))
            g = torch.from_numpy((0.05 * rng.standard_normal((B, DD * DD, H, W))).astype(np.float32))
            data.append((f1, f2, g))
            f.write(struct.pack('7i', R_, B, C, H, W, rows, which))
            for t in (f1, f2, g):
                f.write(t.numpy().tobytes())
    r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout[-2000:], r.stderr[-500:])
    _sanitized(build, [fin, str(tmp_path / 'san.bin')], tmp_path, 'mfma_check')
    raw = np.fromfile(fout, dtype=np.float32)
    pos = 0
    for ((R_, B, C, H, W, rows), which), (f1, f2, g) in zip(cases, data):
        a, b = f1.clone().requires_grad_(), f2.clone().requires_grad_()
        R.corr_naive(a, b, R_).backward(g)
        n = B * C * H * W
        for name, ref in (('gf1', a.grad), ('gf2', b.grad)):
            got = raw[pos:pos + n].reshape(B, C, H, W)
            pos += n
            np.testing.assert_allclose(got, ref.numpy(), rtol=1e-4, atol=1e-5 * float(ref.abs().max()),
                                       err_msg='%s, %s, R=%d [%d,%d,%d,%d] rows %d' % (name, 'corr_mfma2.h' if which else 'corr_mfma.h', R_, B, C, H, W, rows))
    assert pos == raw.size


# (kind, d, B, C, H, W, align_corners, backward mode) -> the kernels the library's dispatch must reach for it (launch-site names, HIP_ON_HOST_TRACE)
_COST_VOLUME_CASES = [
    ((0, 4, 10, 16, 64, 208, 0, 0), ('corr_fwd_ring_kernel', 'corr_bwd_rs_mixed_kernel')),        # level-2 class: all 81 displacements per workgroup; 64x8 + 16x32 tiles
    ((0, 4, 9, 6, 64, 256, 0, 0), ('corr_fwd_ring_kernel', 'corr_bwd_rs_kernel')),                # the same class without a remainder column; a channel group that is not full
    ((0, 4, 5, 8, 32, 208, 0, 0), ('corr_fwd_ringp_kernel', 'corr_bwd_gs_kernel')),               # level-3 class: two channel phases; group-split ring backward
    ((0, 4, 2, 16, 16, 260, 0, 0), ('corr_fwd_ringp_kernel', 'corr_bwd_rs_kernel')),              # level-4 class: four channel phases; row-streamed, 8 channels per item
    ((0, 4, 2, 64, 8, 26, 0, 0), ('corr_fwd_split_kernel', 'corr_bwd_small_kernel')),             # levels 5 / 6: channel slices per wave; whole-map backward
    ((0, 4, 2, 6, 9, 13, 0, 0), ('corr_fwd_generic', 'corr_bwd_small_kernel')),
    ((0, 4, 5, 8, 64, 130, 0, 0), ('corr_fwd_kernel', 'corr_bwd_kernel')),                        # W % 4 != 0: the register-staged tile kernels
    ((0, 4, 17, 4, 64, 130, 0, 0), ('corr_fwd_kernel', 'corr_bwd_kernel')),
    ((0, 8, 3, 16, 32, 104, 0, 1), ('corr_fwd_ring_kernel', 'corr_bwd_rs_kernel')),               # d = 8, fp32 backward on request (the default there is corr_mfma.h: test above)
    ((0, 8, 5, 8, 64, 130, 0, 0), ('corr_fwd_kernel', 'corr_bwd_kernel')),
    ((0, 8, 1, 4, 6, 7, 0, 0), ('corr_fwd_generic', 'corr_bwd_generic')),
    ((0, 1, 2, 5, 9, 37, 0, 0), ('corr_fwd_kernel', 'corr_bwd_kernel')),
    ((0, 2, 2, 5, 9, 37, 0, 0), ('corr_fwd_kernel', 'corr_bwd_kernel')),
    ((0, 3, 1, 4, 7, 9, 0, 0), ('corr_fwd_generic', 'corr_bwd_generic')),                         # a radius no tuned kernel exists for
    # round 6, UNFLOW_CORR_BWD_FP32_NEXT (3): the small-map backward with the gradient rows passing through registers (csrc/corr_small_rows.h; no GPU has run it)
    ((0, 8, 2, 24, 8, 26, 0, 3), ('corr_bwd_smallrows_kernel',)),                                 # level-5 class at d = 8: all lanes pixels, 8 channels per lane, three chunks; the short last row group
    ((0, 8, 3, 21, 4, 13, 0, 3), ('corr_bwd_smallrows_kernel',)),                                 # level-6 class at d = 8: four channel phases x 4 channels, a ragged second chunk
    ((0, 8, 1, 5, 14, 32, 0, 3), ('corr_bwd_smallrows_kernel',)),                                 # two pixel blocks, fewer channels than a chunk
    ((0, 4, 2, 20, 8, 26, 0, 3), ('corr_bwd_smallrows_kernel',)),                                 # d = 4 (three full row groups)
    ((0, 4, 2, 19, 5, 7, 0, 3), ('corr_bwd_smallrows_kernel',)),                                  # a tiny map: dead lanes in every phase
    ((1, 4, 10, 5, 64, 208, 0, 0), ('warp_corr_fwd_kernel',)),                                    # fused warp + cost volume, 81 displacements per workgroup
    ((1, 4, 3, 24, 16, 52, 1, 0), ('warp_corr_fwd_kernel',)),                                     # ... three displacement rows per workgroup, align_corners
    ((1, 4, 2, 5, 23, 72, 0, 0), ('warp_corr_fwd_kernel',)),
]


def test_cost_volume_kernels_run_on_the_host_and_match_the_oracle(tmp_path):
    """csrc/corr.hip and csrc/warp_corr.hip THEMSELVES -- the tile kernels, the LDS-DMA ring kernels with their hand-issued ds_read_b64 /
    ds_write_b64 (plain loads and stores at the same 32-bit LDS addresses on the host), the group-split and row-streamed backward, the
    whole-map backward, the per-element kernels, the fused warp + cost volume -- through the C entry points and the library's own dispatch by
    shape: every case reaches the kernel family it is meant for (launch trace), holds the oracle's `corr_naive` (after `warp_flow`) forward and
    backward at the GPU tests' bars, and runs clean under AddressSanitizer + UBSan (every global and LDS access of every lane inside its
    buffer).  What the host cannot show: the counted waits (every copy has landed when its call returns)."""
    import struct
    clang = '/opt/rocm/lib/llvm/bin/clang++'
    if not os.path.exists(clang):
        import pytest
        pytest.skip('needs clang (vector extensions, __bf16): no ROCm clang++ here')
    csrc = os.path.join(ROOT, 'unopticalflow_amd', 'csrc')
    sources = [os.path.join(csrc, 'corr.hip'), os.path.join(csrc, 'warp_corr.hip'), os.path.join(csrc, 'warp.hip'), os.path.join(ROOT, 'tests', 'host_check', 'corr_check.cpp')]
    common = ['-std=c++20', '-ffp-contract=off', '-DUNFLOW_HOST_CHECK', '-Wno-unknown-attributes', '-Wno-unknown-pragmas', '-Wno-pass-failed',
              '-I', os.path.join(ROOT, 'tests', 'host_check'), '-I', csrc]
    san = ['-O1', '-gline-tables-only', '-fsanitize=address,undefined', '-fno-omit-frame-pointer', '-fno-sanitize-recover=undefined']
    # both programs (plain -O2 and AddressSanitizer + UBSan, see _sanitized) at once, a compiler process per translation unit
    long = _LEVEL == 'all'           # (then both builds: the plain -O2 program's bytes must equal the instrumented -O1 program's)
    plain, sanit = (long or _LEVEL == ''), (long or _LEVEL == '1')
    exe, exe_san = str(tmp_path / 'corr_check'), str(tmp_path / 'corr_check_asan')

    def program(tag, flags, link_flags, out):
        """a compiler process per translation unit, side by side, then the link"""
        import types
        jobs = []
        for src in sources:
            obj = str(tmp_path / ('%s_%s.o' % (tag, os.path.basename(src))))
            jobs.append((obj, subprocess.Popen([clang, *flags, *common, '-x', 'c++', '-c', src, '-o', obj], stderr=subprocess.PIPE, text=True)))
        for obj, pr in jobs:
            err = pr.communicate()[1]
            if pr.returncode != 0:
                return types.SimpleNamespace(returncode=pr.returncode, stderr=err)
        return subprocess.run([clang, *link_flags, '-o', out] + [o for o, _ in jobs], capture_output=True, text=True)
    for want, tag, flags, link_flags, out in ((plain, 'plain', ['-O2'], [], exe), (sanit, 'san', san, ['-fsanitize=address,undefined'], exe_san)):
        if want:
            r = _build_cached([clang, *flags, *common, *sources, '-o', out], builder=lambda: program(tag, flags, link_flags, out))
            assert r.returncode == 0, r.stderr[-3000:]
    rng = np.random.default_rng(5)
    data = []
    fin, fout = str(tmp_path / 'in.bin'), str(tmp_path / 'out.bin')
    with open(fin, 'wb') as f:
        f.write(struct.pack('i', len(_COST_VOLUME_CASES)))
        for (kind, d, B, C, H, W, ac, mode), _ in _COST_VOLUME_CASES:
            DD = 2 * d + 1
            f1, f2 = (torch.from_numpy(rng.uniform(-1, 1, (B, C, H, W)).astype(np.float32)) for _ in range(2))
            # flows that leave the map on every side, sit on pixel centres in places and are smooth elsewhere
            flow = torch.from_numpy((rng.uniform(-1, 1, (B, 2, 1, 1)) * 0.6 * max(H, W) * rng.uniform(0, 1, (B, 1, H, 1)) + rng.uniform(-2, 2, (B, 2, H, W))).astype(np.float32))
            flow[:, :, ::3, ::5] = torch.round(flow[:, :, ::3, ::5])
            g = torch.from_numpy((0.05 * rng.standard_normal((B, DD * DD, H, W))).astype(np.float32))
            data.append((f1, f2, flow, g))
            f.write(struct.pack('8i', kind, d, B, C, H, W, ac, mode))
            for t in ((f1, f2, flow, g) if kind else (f1, f2, g)):
                f.write(t.numpy().tobytes())
    # the program under AddressSanitizer + UBSan (the lanes' stacks are heap blocks switched by hand: no stack-use-after-return tracking), launch trace on
    env = dict(os.environ, ASAN_OPTIONS='detect_stack_use_after_return=0:detect_leaks=0', HIP_ON_HOST_TRACE='1')
    sanitized = subprocess.Popen([exe_san, fin, fout if not plain else str(tmp_path / 'san.bin')], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env) if sanit else None
    if plain:
        r = subprocess.run([exe, fin, fout], capture_output=True, text=True, timeout=3000, env=env)
        assert r.returncode == 0 and 'OK' in r.stdout, (r.stdout[-2000:], r.stderr[-500:])
        se = r.stderr
    if sanit:
        so, se_san = sanitized.communicate(timeout=3000)
        assert sanitized.returncode == 0 and 'OK' in so and 'ERROR' not in se_san and 'runtime error' not in se_san, (so[-1500:], se_san[-3000:])
        se = se if plain else se_san
    launches = [l.split()[1] for l in se.splitlines() if l.startswith('launch ')]
    if long:
        assert open(fout, 'rb').read() == open(str(tmp_path / 'san.bin'), 'rb').read()        # -O1 and -O2, with and without instrumentation: the same bytes
    raw = np.fromfile(fout, dtype=np.float32)
    pos = 0
    at = 0
    for ((kind, d, B, C, H, W, ac, mode), want), (f1, f2, flow, g) in zip(_COST_VOLUME_CASES, data):
        what = 'kind %d d=%d [%d,%d,%d,%d] ac %d' % (kind, d, B, C, H, W, ac)
        # the case's launches: from its first forward kernel to the next case's
        nxt = at + 1
        while nxt < len(launches) and not any(launches[nxt].lstrip('(').startswith(p) for p in ('corr_fwd', 'warp_corr_fwd')):
            nxt += 1
        mine, at = launches[at:nxt], nxt
        for k in want:
            assert any(m.lstrip('(').startswith(k) for m in mine), (what, k, mine)
        a, b, fl = f1.clone().requires_grad_(), f2.clone().requires_grad_(), flow.clone().requires_grad_()
        ref = R.corr_naive(a, R.warp_flow(b, fl, False, bool(ac)) if kind else b, d)
        ref.backward(g)
        DD = 2 * d + 1
        n, ng, nf = B * C * H * W, B * DD * DD * H * W, B * 2 * H * W
        amax = max(float(a.grad.abs().max()), float(b.grad.abs().max()), 1e-6)
        np.testing.assert_allclose(raw[pos:pos + ng].reshape(ref.shape), ref.detach().numpy(), rtol=1e-5, atol=2e-6, err_msg='cv ' + what)
        pos += ng
        np.testing.assert_allclose(raw[pos:pos + n].reshape(a.shape), a.grad.numpy(), rtol=1e-4, atol=max(5e-6, 2e-6 * amax), err_msg='gf1 ' + what)
        pos += n
        np.testing.assert_allclose(raw[pos:pos + n].reshape(a.shape), b.grad.numpy(), rtol=1e-4, atol=max(2e-5, 2e-6 * amax), err_msg='gf2 ' + what)
        pos += n
        if kind:
            scale = max(float(fl.grad.abs().max()), 1e-6)
            np.testing.assert_allclose(raw[pos:pos + nf].reshape(fl.shape), fl.grad.numpy(), rtol=1e-4, atol=2e-5 * scale, err_msg='gflow ' + what)
            pos += nf
    assert pos == raw.size and at == len(launches)
