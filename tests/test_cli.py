"""The reference's command lines (train.py:166-226, test.py:205-255) end to end on the GPU: a prepared-triplet
directory (stacked PNGs + train.txt, what kitti_prepared.py reads) -> train.py -> checkpoints -> resume -> test.py."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

YAML = """cfg_name: 'default'
dataset: 'kitti_depth'
img_hw: [64, 128]
num_scales: 3
num_iterations: 6
w_ssim: 0.85
w_flow_smooth: 10.0
w_flow_consis: 0.01
h_flow_consist_alpha: 3.0
h_flow_consist_beta: 0.05
prepared_base_dir: '%s'
gt_2012_dir: './none'
gt_2015_dir: './none'
"""


def _dataset(root):
    from unopticalflow_amd.evaluation import write_png
    rng = np.random.default_rng(0)
    os.makedirs(os.path.join(root, 'data_s1', 'seq'))
    names = []
    for i, (h, w) in enumerate([(75, 248), (74, 244), (80, 260), (75, 248), (90, 300), (64, 128)]):
        base = rng.integers(0, 256, (h + 8, w + 8, 3), dtype=np.uint8)
        frames = [base[4 + dy:4 + dy + h, 4 + dx:4 + dx + w] for dy, dx in ((0, -3), (0, 0), (1, 3))]   # shifted views
        write_png(os.path.join(root, 'data_s1', 'seq', '%d.png' % i), np.concatenate(frames, 0))
        names.append('seq/%d.png seq/%d_cam.txt' % (i, i))
    with open(os.path.join(root, 'data_s1', 'train.txt'), 'w') as f:
        f.write('\n'.join(names) + '\n')


@pytest.mark.parametrize('host_input', [0, 1])
def test_train_resume_and_test_cli(tmp_path, host_input, capsys):
    from unopticalflow_amd import train as train_cli, test as test_cli
    root = str(tmp_path)
    _dataset(root)
    cfg = os.path.join(root, 'cfg.yaml')
    with open(cfg, 'w') as f:
        f.write(YAML % root)
    common = ['-c', cfg, '--gpu', '0', '--mode', 'flow', '--model_dir', os.path.join(root, 'models'), '--batch_size', '2',
              '--num_workers', '0', '--log_interval', '1', '--save_interval', '3', '--miopen_find', '0',
              '--host_input', str(host_input)]
    cwd = os.getcwd()
    os.chdir(root)
    try:
        trainer = train_cli.main(common)
        out = capsys.readouterr().out
        assert 'iter: 5, loss_pixel:' in out and 'pairs/s' in out            # Visualizer.print_loss line + rate
        mdir = os.path.join(root, 'models', 'flow')
        for name in ('iter_2.pth', 'iter_5.pth', 'last.pth', 'config.pkl', 'cfg.yaml'):
            assert os.path.exists(os.path.join(mdir, name)), name
        ck = torch.load(os.path.join(mdir, 'last.pth'), map_location='cpu')
        assert set(ck) == {'iteration', 'model_state_dict', 'optimizer_state_dict'} and ck['iteration'] == 5
        assert len(ck['model_state_dict']) == 98 and 'pwc_model.dc_conv7.bias' in ck['model_state_dict']
        for k, v in trainer.model.state_dict().items():
            assert torch.equal(v.cpu(), ck['model_state_dict'][k]), k
        # resume from iter_2 and run to the end again (train.py:42-46)
        trainer2 = train_cli.main(common + ['--resume', '--iter_start', '2'])
        assert 'iter: 5, loss_pixel:' in capsys.readouterr().out
        assert all(torch.isfinite(p).all() for p in trainer2.model.parameters())
        # --flow_pretrained_model (train.py:47-61): weights only, fresh optimizer, DataParallel-prefixed keys accepted
        ck['model_state_dict'] = {'module.' + k: v for k, v in ck['model_state_dict'].items()}
        torch.save(ck, os.path.join(root, 'published.pth'))
        trainer3 = train_cli.main(common + ['--flow_pretrained_model', os.path.join(root, 'published.pth'), '--num_iterations', '2'])
        assert 'Load Flow Pretrained Model' in capsys.readouterr().out and trainer3.iteration == 2
        # test.py on the saved checkpoint (synthetic task: no dataset needed)
        res = test_cli.main(['-c', cfg, '--gpu', '0', '--mode', 'flow', '--task', 'synthetic_flow',
                             '--pretrained_model', os.path.join(mdir, 'last.pth')])
        assert res is not None
    finally:
        os.chdir(cwd)


def test_bench_starts_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` with NO launcher in front (the driver's command form; the reference takes its GPU list on one
    command line, train.py:208-214): the parent starts two fresh ranks, relays rank 0's JSON as its last stdout line, exits 0.
    Rehearsal on this one-GPU box: both ranks share device 0 over gloo (UNFLOW_BENCH_ONE_GPU=1) -- the numbers mean nothing,
    the plumbing (self-launch, replayed step with one all-reduce, barriers, MAX over ranks, rank spread) is the real one."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UNFLOW_BENCH_ONE_GPU='1', UNFLOW_MIOPEN_FIND='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '2',
                        '--hw', '64', '128', '--batch', '2', '--no-cpu-baseline'], env=env, cwd=str(tmp_path),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 2 and line['steps'] == 3 and line['value'] > 0 and line['config']['parallelism'] == 'dp2'
    assert line['step_mode'].startswith('hipGraph replay') and line['rank_spread']['max_over_ranks']['step_ms_median'] > 0


@pytest.mark.skipif(os.environ.get('UNFLOW_RUN_EIGHT_RANKS') != '1',
                    reason='nine HIP contexts on one GPU (eight ranks + this process): run on its own, `bash tools/gpu_r6.sh ranks8` '
                           '(once in three runs inside the whole suite the runtime aborted THIS process while the ranks started)')
@pytest.mark.gpu
def test_bench_eight_ranks_rehearsal_and_a_killed_rank(tmp_path):
    """nobody can measure eight GPUs here, so the eight-rank form of the driver's command is rehearsed on the one-GPU box
    (eight self-launched ranks sharing device 0 over gloo): port / rendezvous, eight private find-db copies, the OMP_NUM_THREADS
    split, rank_spread, one JSON line with n_gpus 8 -- and a rank that is killed ends the whole job non-zero within the grace period
    (the reference's one-command form, train.py:208-214)."""
    import json
    import signal
    import subprocess
    import sys
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, UNFLOW_BENCH_ONE_GPU='1', UNFLOW_MIOPEN_FIND='0')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT', 'OMP_NUM_THREADS'):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, 'bench.py'), '--gpus', '8', '--steps', '3', '--warmup', '3', '--hw', '64', '128', '--batch', '2',
           '--no-cpu-baseline']
    r = subprocess.run(cmd, env=env, cwd=str(tmp_path), capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line['n_gpus'] == 8 and line['config']['parallelism'] == 'dp8' and line['config']['global_batch'] == 16 and line['value'] > 0
    assert line['scaling'] == 'weak' and isinstance(line['cpu_baseline'], str) and 'N=1' in line['cpu_baseline']
    assert line['rank_spread']['max_over_ranks']['step_ms_median'] >= line['rank_spread']['min_over_ranks']['step_ms_median'] > 0
    # one rank killed in the middle of a (long) run: the launcher ends every other rank and reports failure
    p = subprocess.Popen(cmd[:4] + ['--steps', '2000', '--warmup', '3'] + cmd[8:], env=env, cwd=str(tmp_path), stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True)
    victim = None
    t0 = time.monotonic()
    while time.monotonic() - t0 < 600 and victim is None:              # a rank process of OUR launcher: a child of p with RANK=3
        time.sleep(1.0)
        for d in os.listdir('/proc'):
            if not d.isdigit():
                continue
            try:
                stat = open('/proc/%s/stat' % d).read().rsplit(')', 1)[1].split()
                if int(stat[1]) != p.pid:
                    continue
                if b'RANK=3' in open('/proc/%s/environ' % d, 'rb').read().split(b'\0'):
                    victim = int(d)
            except (OSError, IndexError, ValueError):
                continue
    assert victim is not None, 'rank 3 never appeared'
    time.sleep(20.0)                                                   # (let the ranks get past the rendezvous and into the steps)
    os.kill(victim, signal.SIGKILL)                                    # an exact PID, found by parent + environment
    t1 = time.monotonic()
    rc = p.wait(timeout=120)
    assert rc != 0 and time.monotonic() - t1 < 60, rc
