"""MIOpen solver selection for the convolution stacks.

The flow step is convolution-bound (~24 of ~29 ms on an MI355X at 832x256, B=8, fp32).  MIOpen's
default (immediate-mode heuristic) picks are ~5 % slower than the measured best per layer, so this
package ships the *find-db* MIOpen wrote after an exhaustive find of exactly the step's conv
configurations on an MI355X (``miopen_db/*.ufdb.txt`` / ``*.udb.txt``: plain-text lists of
solver -> measured ms per conv config, produced by ``tools/gpu_find.sh``), and points MIOpen at it.

    enable_miopen_tuning()      # before the first convolution

sets MIOPEN_USER_DB_PATH (unless the user already did) and ``torch.backends.cudnn.benchmark``.
Configs that are not in the shipped db (other resolutions / batch sizes) make MIOpen run its find
once (minutes on first use) and append to the db.
"""
import os

import torch

DB_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'miopen_db')


def enable_miopen_tuning(benchmark=True):
    os.environ.setdefault('MIOPEN_USER_DB_PATH', DB_DIR)
    torch.backends.cudnn.benchmark = bool(benchmark)
    return os.environ['MIOPEN_USER_DB_PATH']
