"""Start one process per GPU from a single command line.

The reference takes its GPU list on ONE command (``python train.py --gpu 0,1,2,3 --multi_gpu``, train.py:36-37,208-214:
single-process DataParallel).  Here the data-parallel job is one process per GPU over RCCL; ``spawn_ranks`` is what lets
``python bench.py --gpus N`` and ``python -m unopticalflow_amd.train --gpu 0,..,N-1 --multi_gpu`` keep that one-command form
without ``torch.distributed.run`` in front: the parent starts N FRESH children (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
their environment), relays rank 0's stdout as its own and returns the exit code of the rank that failed first (0: all succeeded).

The parent never touches the GPU: this module imports nothing but the standard library, and a caller must invoke it
before it initialises HIP (a process that has initialised the GPU must not exec or fork workers that use it).
"""
import os
import signal
import socket
import subprocess
import sys
import threading
import time


def free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launched_by_torchrun(env=None):
    """True inside a process that a launcher (torch.distributed.run or ``spawn_ranks``) has already given a rank."""
    env = os.environ if env is None else env
    return 'RANK' in env and 'WORLD_SIZE' in env


def rank_env(rank, world, port, base=None):
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')            # the host driver only does dmabuf IPC (RCCL needs it)
    if 'OMP_NUM_THREADS' not in env:                              # as torch.distributed.run: do not let N ranks each spin up every core
        try:
            cores = len(os.sched_getaffinity(0))
        except AttributeError:
            cores = os.cpu_count() or 1
        env['OMP_NUM_THREADS'] = str(max(1, cores // world))
    return env


def _libc_prctl():
    """libc's prctl, resolved ONCE at import: the hook below runs in the forked child between fork and exec, where an import or a dlopen could
    deadlock on a lock some other thread of the parent held at the fork (pump threads of an earlier spawn_ranks call, torch's pools)."""
    try:
        import ctypes
        return ctypes.CDLL('libc.so.6', use_errno=True).prctl
    except Exception:                                              # not Linux / no libc: the signal forwarding still covers SIGTERM / SIGHUP
        return None


_PRCTL = _libc_prctl()
_SIGKILL = int(signal.SIGKILL)


def _die_with_parent():
    """preexec hook of a rank: SIGKILL when the launcher dies -- a launcher killed with SIGKILL (OOM killer, ``timeout -s KILL``) cannot
    forward anything, and orphaned ranks would keep the GPUs and the port.  One call into an already-resolved C function: no import, no
    allocation.  PR_SET_PDEATHSIG follows the forking THREAD, which is why spawn_ranks insists on the main thread."""
    if _PRCTL is not None:
        _PRCTL(1, _SIGKILL, 0, 0, 0)                               # PR_SET_PDEATHSIG


def spawn_ranks(argv, world, poll_s=0.2, grace_s=10.0, out=None, err=None):
    """Run ``argv`` (a full command, e.g. ``[sys.executable, 'bench.py', '--gpus', '8']``) as ``world`` ranks.

    Rank 0's stdout is forwarded line by line to ``out`` (default: this process's stdout) -- a rank-0 program that prints its
    result last keeps it last; the other ranks' stdout goes to ``err`` (stderr), every rank's stderr is inherited.  When a rank
    fails, the others (which would wait for it in a collective until the watchdog fires) are terminated by PID.  Returns 0 when
    every rank succeeded, else the exit code of the rank that failed first (128 + signal for a killed rank).

    A SIGTERM / SIGHUP / Ctrl-C to the launcher (scheduler stop, ``timeout ... python bench.py --gpus N``, closed terminal) is
    forwarded to the exact child PIDs; ranks still alive ``grace_s`` later are killed; the launcher then returns 128 + signal
    (Ctrl-C: re-raises KeyboardInterrupt) -- no rank outlives it, and a launcher that is SIGKILLed takes its ranks with it
    (PR_SET_PDEATHSIG)."""
    if world < 1:
        raise ValueError('world must be >= 1')
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port()
    stop = {'sig': None}

    def on_signal(signum, frame):
        stop['sig'] = signum
    old = {}
    # signal.signal is main-thread only, and PR_SET_PDEATHSIG is tied to the forking THREAD (the ranks would be killed when a helper thread that
    # called us returns): from another thread the defaults stay and the ranks get no parent-death signal
    on_main = threading.current_thread() is threading.main_thread()
    if on_main:
        for sg in (signal.SIGTERM, signal.SIGHUP):
            old[sg] = signal.signal(sg, on_signal)
    procs = []
    try:
        for r in range(world):
            procs.append(subprocess.Popen(list(argv), env=rank_env(r, world, port), stdout=subprocess.PIPE,
                                          stderr=None, text=True, bufsize=1, preexec_fn=_die_with_parent if on_main else None))

        def pump(p, sink):
            for line in p.stdout:
                sink.write(line)
                sink.flush()
        threads = [threading.Thread(target=pump, args=(p, out if r == 0 else err), daemon=True) for r, p in enumerate(procs)]
        for t in threads:
            t.start()

        def signal_all(sg):
            for p in procs:                                       # exact PIDs of our own children, never a pattern
                if p.poll() is None:
                    p.send_signal(sg)
        failed, first_bad, forwarded, interrupted = None, 0, None, False
        while True:
            try:
                codes = [p.poll() for p in procs]
                if stop['sig'] is not None and forwarded is None:
                    forwarded = stop['sig']
                    signal_all(forwarded)
                    failed = failed if failed is not None else time.monotonic()
                bad = [c for c in codes if c not in (None, 0)]
                if bad and failed is None:
                    failed, first_bad = time.monotonic(), (128 - bad[0] if bad[0] < 0 else bad[0])
                    signal_all(signal.SIGTERM)
                if all(c is not None for c in codes):
                    break
                if failed is not None and time.monotonic() - failed > grace_s:
                    for p in procs:
                        if p.poll() is None:
                            p.kill()
                time.sleep(poll_s)
            except KeyboardInterrupt:                              # (the terminal sent SIGINT to the ranks too; make sure, then wait like above)
                if interrupted:
                    for p in procs:
                        if p.poll() is None:
                            p.kill()
                interrupted = True
                signal_all(signal.SIGTERM)
                failed = failed if failed is not None else time.monotonic()
        for t in threads:
            t.join(timeout=5.0)
    finally:
        for p in procs:                                           # (an exception on the way: nothing of ours is left behind)
            if p.poll() is None:
                p.kill()
        for sg, h in old.items():
            signal.signal(sg, h if h is not None else signal.SIG_DFL)      # (None: the previous handler was installed from C)
    if interrupted:
        raise KeyboardInterrupt
    if forwarded is not None:
        err.write('launch: stopped by signal %d\n' % forwarded)
        return 128 + int(forwarded)
    worst = 0
    for r, p in enumerate(procs):
        c = p.returncode
        c = 128 - c if c < 0 else c
        if c != 0:
            err.write('launch: rank %d exited with code %d\n' % (r, c))
        worst = max(worst, c)
    return first_bad or worst                                     # the rank that failed first, not the SIGTERM we answered with
