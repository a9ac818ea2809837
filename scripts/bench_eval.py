"""Inference latency / throughput of the detector (eval mode, test-time proposal counts): BASELINE
configs[0] (1 pair, the reference's CPU-runnable case: 2.70 s/pair measured on 8 host cores,
SURVEY 8d) and a batch of 8 pairs.  Not the headline metric (that is the training step)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from ait_amd import tuning
from ait_amd.config import cfg
tuning.use_tuned_miopen_db(0)
dev = torch.device("cuda:0")
m = bench.build_model(300, dev).eval()
for bs in (1, 8):
    for P in (128, 300):
        cfg.TEST.RPN_POST_NMS_TOP_N = P
        np.random.seed(3)
        batch = bench.synth_batch(bs, 1000, dev)
        with torch.no_grad():
            for _ in range(3): out = m(*batch)
            torch.cuda.synchronize(); t = time.time()
            n = 10
            for _ in range(n): out = m(*batch)
            torch.cuda.synchronize(); dt = (time.time() - t) / n
        print("eval forward bs=%d P=%d: %.1f ms/batch  %.1f pairs/s  (rois %s)" % (bs, P, dt * 1e3, bs / dt, tuple(out[0].shape)))
