"""MIOpen algorithm selection for the PyTorch-side convolutions (ResNet trunk, layer4 head, SKNet,
RPN), which SURVEY.md section 2 keeps on PyTorch-ROCm.

`ait_amd/miopen_db/` holds MIOpen's user find-db / perf-db recorded on an MI355X (gfx950, 256 CU)
for exactly the convolution shapes of the bench workloads cfg2 .. cfg5 (scripts/exp_miopen_db.py: one run per
configuration with torch.backends.cudnn.benchmark = True; cfg5: 122 instead of 148 ms/step, cfg3: 86.5 instead of
96 ms/step against MIOpen's immediate mode).  With the db in place the measured-best
solver per shape is picked immediately (first step 47 s instead of 668 s) and the training step is
~7 % faster than with MIOpen's immediate-mode heuristics (103.9-107 vs 112.9 ms at bs=4, P=300).
"""
import os
import shutil
import tempfile

_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "miopen_db")


def use_tuned_miopen_db(rank=0):
    """Point MIOpen at a private, writable copy of the committed db and turn on find mode.
    Must run before the first convolution.  Returns True if enabled."""
    if os.environ.get("AIT_MIOPEN_FIND", "1") == "0":
        return False
    files = [f for f in os.listdir(_DB)] if os.path.isdir(_DB) else []
    if not any(f.startswith("gfx950") and f.endswith(".ufdb.txt") for f in files):
        return False
    dst = os.path.join(tempfile.gettempdir(), "ait_miopen_db_%d_rank%d" % (os.getuid(), rank))
    os.makedirs(dst, exist_ok=True)
    for f in files:
        if not os.path.exists(os.path.join(dst, f)):
            shutil.copy(os.path.join(_DB, f), os.path.join(dst, f))
    os.environ["MIOPEN_USER_DB_PATH"] = dst
    os.environ.setdefault("MIOPEN_CUSTOM_CACHE_DIR", os.path.join(dst, "cache"))
    import torch
    torch.backends.cudnn.benchmark = True
    return True
