"""A/B of the bf16 configuration's proposal tail: `python scripts/ab_cfg5_tail.py miopen|library [bench.py arguments]` runs
bench.py with faster_rcnn._TAIL_BF16_MIOPEN set accordingly (miopen: the module composition on MIOpen's bf16 convolutions;
library: ait_tail_* on bf16 storage)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ait_amd.faster_rcnn as fr
import bench

if __name__ == "__main__":
    fr._TAIL_BF16_MIOPEN = sys.argv[1] == "miopen"
    sys.argv = ["bench.py"] + sys.argv[2:]
    bench.main()
