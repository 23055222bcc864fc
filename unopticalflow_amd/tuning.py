"""MIOpen solver selection for the convolution stacks.

The flow step is convolution-bound (~24 of ~29 ms on an MI355X at 832x256, B=8, fp32).  MIOpen's
default (immediate-mode heuristic) picks are ~5 % slower than the measured best per layer, so this
package ships the *find-db* MIOpen wrote after an exhaustive find of exactly the step's conv
configurations on an MI355X (``miopen_db/*.ufdb.txt`` / ``*.udb.txt``: plain-text lists of
solver -> measured ms per conv config, produced by ``tools/gpu_find.sh``), and points MIOpen at it.

    enable_miopen_tuning()      # before the first convolution

sets MIOPEN_USER_DB_PATH (unless the user already did) and ``torch.backends.cudnn.benchmark``.
Configs that are not in the shipped db (other resolutions / batch sizes) make MIOpen run its find
once (20+ minutes on first use for a full-size resolution: ~140 conv configs) and append to the db.  With several ranks per node every process works on a
private copy of the shipped files (removed at exit), so no two processes ever append to the same file.
"""
import atexit
import os
import shutil
import tempfile

import torch

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'miopen_db')


def enable_miopen_tuning(benchmark=True):
    ours = 'MIOPEN_USER_DB_PATH' not in os.environ          # a user-provided db path is the user's business
    if ours:
        path = DB_DIR
        if int(os.environ.get('WORLD_SIZE', '1')) > 1:
            path = tempfile.mkdtemp(prefix='unflow_miopen_rank%s_' % os.environ.get('RANK', '0'))
            for name in os.listdir(DB_DIR):
                shutil.copy(os.path.join(DB_DIR, name), path)
            atexit.register(shutil.rmtree, path, ignore_errors=True)
        os.environ['MIOPEN_USER_DB_PATH'] = path
    # The db files are keyed by architecture + CU count ("gfx950100" = gfx950, 0x100 CUs).  On a device the shipped db
    # does not cover (another partition mode / SKU) benchmark mode would start an exhaustive find of ~140 configs
    # (>10 min): stay on MIOpen's immediate-mode heuristics there (~5 % slower) unless the user insists.
    if ours and benchmark and torch.cuda.is_available() and os.environ.get('UNFLOW_MIOPEN_FORCE_FIND') != '1':
        prop = torch.cuda.get_device_properties(torch.cuda.current_device())
        key = '%s%x' % (prop.gcnArchName.split(':')[0], prop.multi_processor_count)
        if not any(n.startswith(key) for n in os.listdir(os.environ['MIOPEN_USER_DB_PATH'])):
            benchmark = False
    torch.backends.cudnn.benchmark = bool(benchmark)
    return os.environ['MIOPEN_USER_DB_PATH']
