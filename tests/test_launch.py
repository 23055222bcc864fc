"""unopticalflow_amd.launch: one command line starts one process per GPU (the reference's `--gpu 0,1,.. --multi_gpu` form,
train.py:208-214) -- CPU checks of the launcher itself; the GPU rehearsal of `python bench.py --gpus 2` is in test_cli.py."""
import io
import os
import sys
import textwrap
import time

from unopticalflow_amd.launch import launched_by_torchrun, rank_env, spawn_ranks

WORKER = textwrap.dedent('''
    import os, sys, time
    r, w = os.environ['RANK'], os.environ['WORLD_SIZE']
    assert os.environ['LOCAL_RANK'] == r and os.environ['MASTER_ADDR'] == '127.0.0.1' and int(os.environ['MASTER_PORT']) > 0
    assert os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    print('rank %s of %s' % (r, w), flush=True)
    if len(sys.argv) > 1 and r == sys.argv[1]:
        sys.exit(3)
    time.sleep(float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
    print('{"rank": %s}' % r, flush=True)
''')


def test_ranks_get_their_environment_and_rank0_owns_stdout(tmp_path):
    f = tmp_path / 'w.py'
    f.write_text(WORKER)
    out, err = io.StringIO(), io.StringIO()
    assert spawn_ranks([sys.executable, str(f)], 3, out=out, err=err) == 0
    assert out.getvalue().splitlines() == ['rank 0 of 3', '{"rank": 0}']          # rank 0's last line stays the last line
    assert sorted(l for l in err.getvalue().splitlines() if l.startswith('{')) == ['{"rank": 1}', '{"rank": 2}']


def test_a_failed_rank_ends_the_job_with_its_code(tmp_path):
    f = tmp_path / 'w.py'
    f.write_text(WORKER)
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    rc = spawn_ranks([sys.executable, str(f), '1', '60'], 3, out=out, err=err)      # rank 1 exits 3, the others would sleep 60 s
    assert rc == 3 and time.monotonic() - t0 < 20
    assert 'rank 1 exited with code 3' in err.getvalue() and '{"rank": 0}' not in out.getvalue()


def test_rank_env_and_detection():
    e = rank_env(2, 4, 1234, base={'PATH': os.environ.get('PATH', '')})
    assert (e['RANK'], e['LOCAL_RANK'], e['WORLD_SIZE'], e['MASTER_PORT']) == ('2', '2', '4', '1234')
    assert int(e['OMP_NUM_THREADS']) >= 1
    assert launched_by_torchrun(e) and not launched_by_torchrun({'PATH': ''})
    assert rank_env(0, 2, 1, base={'OMP_NUM_THREADS': '7'})['OMP_NUM_THREADS'] == '7'


def test_bench_parent_does_not_import_torch_before_launching():
    """`python bench.py --gpus N` decides to self-launch before `import torch` (the parent must never initialise HIP)."""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'bench.py')).read()
    assert src.index('self_launch_if_needed()\n\nimport torch') > 0
    assert src.index('def self_launch_if_needed') < src.index('\nimport torch')


def test_train_cli_starts_one_rank_per_listed_gpu(tmp_path, monkeypatch):
    """`python -m unopticalflow_amd.train --gpu 0,1,2 --multi_gpu` with no launcher in front (the reference's command form,
    train.py:198-214): the parent hands its own command line to spawn_ranks with one rank per listed GPU and exits with its code --
    before it creates directories, touches a GPU or reads a dataset; behind a launcher nothing is started again."""
    import pytest
    from unopticalflow_amd import launch, train as train_cli
    cfg = tmp_path / 'k.yaml'
    cfg.write_text("dataset: 'kitti_depth'\nimg_hw: [64, 128]\nnum_scales: 3\nnum_iterations: 4\n")
    seen = {}

    def fake_spawn(cmd, world):
        seen['cmd'], seen['world'] = cmd, world
        return 7
    monkeypatch.setattr(launch, 'spawn_ranks', fake_spawn)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK'):
        monkeypatch.delenv(k, raising=False)
    argv = ['-c', str(cfg), '--gpu', '0,1,2', '--multi_gpu', '--synthetic', '--model_dir', str(tmp_path / 'm')]
    with pytest.raises(SystemExit) as e:
        train_cli.main(argv)
    assert e.value.code == 7 and seen['world'] == 3
    assert seen['cmd'][1:3] == ['-m', 'unopticalflow_amd.train'] and seen['cmd'][3:] == argv
    assert not (tmp_path / 'm').exists()
    # the reference's own consistency check between --gpu and --multi_gpu (train.py:208-210) still comes first
    with pytest.raises(ValueError):
        train_cli.main(['-c', str(cfg), '--gpu', '0,1', '--synthetic'])
    # behind a launcher (RANK / WORLD_SIZE set) the world size must match the list: no second launch, a clear error otherwise
    monkeypatch.setenv('RANK', '0'); monkeypatch.setenv('WORLD_SIZE', '2'); monkeypatch.setenv('LOCAL_RANK', '0')
    seen.clear()
    with pytest.raises(ValueError, match='launcher started 2'):
        train_cli.main(argv)
    assert not seen


def _pid_alive(pid):
    try:
        os.kill(pid, 0)
    except ProcessLookupError:
        return False
    try:                                                   # (a zombie still answers kill 0: look at its state)
        return open('/proc/%d/stat' % pid).read().rsplit(')', 1)[1].split()[0] != 'Z'
    except OSError:
        return False


def _start_launcher(tmp_path, sleep_s=120):
    import subprocess
    w = tmp_path / 'w.py'
    w.write_text(textwrap.dedent('''
        import os, sys, time
        open(os.path.join(%r, 'pid%%s' %% os.environ['RANK']), 'w').write(str(os.getpid()))
        time.sleep(%d)
    ''' % (str(tmp_path), sleep_s)))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    parent = subprocess.Popen([sys.executable, '-c',
                               'import sys; sys.path.insert(0, %r)\n'
                               'from unopticalflow_amd.launch import spawn_ranks\n'
                               'sys.exit(spawn_ranks([sys.executable, %r], 2, grace_s=3.0))' % (root, str(w))])
    t0 = time.monotonic()
    while time.monotonic() - t0 < 30 and not all((tmp_path / ('pid%d' % r)).exists() and (tmp_path / ('pid%d' % r)).read_text() for r in (0, 1)):
        time.sleep(0.05)
    pids = [int((tmp_path / ('pid%d' % r)).read_text()) for r in (0, 1)]
    assert all(_pid_alive(p) for p in pids)
    return parent, pids


def test_sigterm_to_the_launcher_stops_every_rank(tmp_path):
    """`timeout ... python bench.py --gpus N` / a scheduler stop sends SIGTERM to the PARENT only: it is forwarded to the
    exact child PIDs, the launcher waits for them and exits 128 + 15 -- no rank is left holding a GPU or the rendezvous port."""
    import signal
    parent, pids = _start_launcher(tmp_path)
    parent.send_signal(signal.SIGTERM)
    assert parent.wait(timeout=20) == 128 + signal.SIGTERM
    assert not any(_pid_alive(p) for p in pids)


def test_a_killed_launcher_takes_its_ranks_with_it(tmp_path):
    """SIGKILL cannot be forwarded: the ranks carry PR_SET_PDEATHSIG and die with the launcher."""
    parent, pids = _start_launcher(tmp_path)
    parent.kill()
    parent.wait(timeout=20)
    t0 = time.monotonic()
    while time.monotonic() - t0 < 10 and any(_pid_alive(p) for p in pids):
        time.sleep(0.1)
    assert not any(_pid_alive(p) for p in pids)
