"""bench.py's OWN code on CPU tensors over the host-executed kernels -- TEST INFRASTRUCTURE (tests/rehearsal.py + tests/hostexec.py).

    python tests/bench_on_host.py --steps 2 --warmup 1 --batch 1 --hw 64 128 --graph 0 [bench.py's other flags]

What it shows: the plumbing of the bench line (argument handling, the step loop, kernel-timer bookkeeping, the roofline / cpu_baseline objects,
one JSON line last on stdout).  What it cannot: any number -- host timers report a made-up microsecond per timed slot."""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'tests')]
import hostexec  # noqa: E402
import rehearsal  # noqa: E402

rehearsal.install()
from unopticalflow_amd import ops  # noqa: E402

sys.argv = [os.path.join(ROOT, 'bench.py')] + sys.argv[1:]
with hostexec.patched(ops):
    runpy.run_path(os.path.join(ROOT, 'bench.py'), run_name='__main__')
