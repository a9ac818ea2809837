import sys, os
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import bench
mode = sys.argv[1]
dev = torch.device("cuda:0")
model = bench.build_model(64, dev)
batch = bench.synth_batch(2, 1000, dev, max_gt=20)
np.random.seed(3)
if mode == "pre":
    q = batch[1]
    f = model._query_trunk(q)
    print("pre-capture ok", f.shape, flush=True)
if mode == "img_first":
    g = model.RCNN_base(batch[0])[0]
    print("image trunk ok", flush=True)
    f = model._query_trunk(batch[1])
    print("capture after image trunk ok", flush=True)
    sys.exit(0)
if mode == "anchor_first":
    from ait_amd.faster_rcnn import _c4_size
    model.RCNN_rpn.RPN_anchor_target.begin(batch[3], batch[2], *_c4_size(batch[0].size(2), batch[0].size(3)), im_hw_hint=(batch[0].size(2), batch[0].size(3)))
    print("anchor begin ok", flush=True)
    f = model._query_trunk(batch[1])
    print("capture after anchor begin ok", flush=True)
    sys.exit(0)
out = model(*batch)
bench.total_cost(out).backward()
torch.cuda.synchronize()
print("step ok", mode, flush=True)
