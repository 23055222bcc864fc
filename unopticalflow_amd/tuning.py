"""MIOpen solver selection for the convolution stacks.

The flow step is convolution-bound (~20 of ~26 ms on an MI355X at 832x256, B=8, fp32).  MIOpen's
default (immediate-mode heuristic) picks are ~5 % slower than the measured best per layer, so this
package ships the *find-db* MIOpen wrote after an exhaustive find of exactly the step's conv
configurations on an MI355X (``miopen_db/*.ufdb.txt`` / ``*.udb.txt``: plain-text lists of
solver -> measured ms per conv config, produced by ``tools/gpu_find.sh``), and points MIOpen at it.

    enable_miopen_tuning()      # before the first convolution

* Every process works on a PRIVATE copy of the shipped files (a temp dir, removed at exit): MIOpen appends to its
  user db whenever it meets a conv config that is not in it (another resolution or batch size, a ragged last
  batch), and neither the package directory (possibly read-only, git-tracked) nor a file shared with another
  process may be written.  ``tools/gpu_find.sh`` is the only writer of ``miopen_db/``.
* The db file names carry the device ("gfx950100" = gfx950, 0x100 CUs) and the MIOpen build
  ("HIP.3_5_0_<tweak>").  Benchmark (find) mode is switched on only when the shipped files match BOTH for the
  MIOpen library this process has loaded; otherwise MIOpen would ignore them and start an exhaustive find of ~140
  configs (20+ minutes) on the first iterations, so the process stays on immediate-mode heuristics (~5 % slower)
  unless the user sets ``UNFLOW_MIOPEN_FORCE_FIND=1``.  A user-provided ``MIOPEN_USER_DB_PATH`` is left alone.
"""
import atexit
import mmap
import os
import re
import shutil
import tempfile

import torch

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'miopen_db')


def loaded_miopen_db_tag():
    """'HIP.<major>_<minor>_<patch>_<tweak>' of the MIOpen library mapped into this process (the middle part of its
    user-db file names), or None when it cannot be determined."""
    path = None
    try:
        with open('/proc/self/maps') as f:
            for ln in f:
                if 'libMIOpen' in ln:
                    path = ln.split()[-1]
                    break
    except OSError:
        return None
    if path is None or not os.path.exists(path):
        return None
    try:
        with open(path, 'rb') as f:
            m = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
            try:
                r = re.search(rb'HIP\.\d+_\d+_\d+_[0-9A-Za-z\-]+', m)
                tag = r.group(0).decode() if r else None      # (a match object dies with the map)
            finally:
                m.close()
    except (OSError, ValueError):
        return None
    return tag


def shipped_db_matches(device_key, miopen_tag, names=None):
    """True when a shipped db file is named exactly as this device + MIOpen build would name it."""
    if device_key is None or miopen_tag is None:
        return False
    names = os.listdir(DB_DIR) if names is None else names
    want = '%s.%s.' % (device_key, miopen_tag)
    return any(n.startswith(want) for n in names)


def private_db_copy():
    path = tempfile.mkdtemp(prefix='unflow_miopen_%d_' % os.getpid())
    for name in os.listdir(DB_DIR):
        shutil.copy(os.path.join(DB_DIR, name), path)
    atexit.register(shutil.rmtree, path, ignore_errors=True)
    return path


_state = {'measured_picks': False}


def default_channels_last():
    """THE default of ``cfg.channels_last`` (Model_flow, train.py, test.py, bench.py all ask here): NHWC conv stacks pay
    off with MIOpen's MEASURED solver picks (the shipped find-db in benchmark mode: fp32 26.1 -> 25.3 ms, bf16 15.9 ->
    14.1 ms per step) and lose with its immediate-mode heuristics (29.2 vs 27.3 ms), so the default is True exactly when
    ``enable_miopen_tuning()`` has switched find mode on for a db that matches this device and MIOpen build."""
    return bool(_state['measured_picks'] and torch.backends.cudnn.benchmark)


def enable_miopen_tuning(benchmark=True):
    ours = 'MIOPEN_USER_DB_PATH' not in os.environ          # a user-provided db path is the user's business
    if ours:
        os.environ['MIOPEN_USER_DB_PATH'] = private_db_copy()
    if ours and benchmark and os.environ.get('UNFLOW_MIOPEN_FORCE_FIND') != '1':
        key = None
        if torch.cuda.is_available():
            prop = torch.cuda.get_device_properties(torch.cuda.current_device())
            key = '%s%x' % (prop.gcnArchName.split(':')[0], prop.multi_processor_count)
        if not shipped_db_matches(key, loaded_miopen_db_tag()):
            benchmark = False
    torch.backends.cudnn.benchmark = bool(benchmark)
    _state['measured_picks'] = bool(benchmark)
    return os.environ['MIOPEN_USER_DB_PATH']
