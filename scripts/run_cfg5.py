"""Sanity run at the largest BASELINE configuration (cfg 5: COCO variant, ResNet101, 8 pairs per GPU,
512 proposals, bf16 AIT GEMMs): one training step runs, is finite, and fits comfortably in HBM.
(MIOpen immediate mode: AIT_MIOPEN_FIND=0, the recorded find-db is for the cfg-2 shapes.)"""
import os, sys, time
os.environ.setdefault("AIT_MIOPEN_FIND", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from ait_amd import ops
from ait_amd.config import cfg, cfg_from_list
from ait_amd.faster_rcnn import resnet_coco
import bench
bs, P = 8, 512
cfg_from_list(['ANCHOR_SCALES', [4, 8, 16, 32], 'MAX_NUM_GT_BOXES', 50, 'TRAIN.BATCH_SIZE', P])
ops.set_matmul_dtype(sys.argv[1] if len(sys.argv) > 1 else "bf16")
torch.manual_seed(0); np.random.seed(3)
m = resnet_coco(('bg', 'fg'), 101, pretrained=False, class_agnostic=True, num_K=3); m.create_architecture()
m = m.cuda().train()
opt = torch.optim.SGD([p for p in m.parameters() if p.requires_grad], lr=1e-3, momentum=0.9)
ins = bench.synth_batch(bs, 5, torch.device("cuda:0"), max_gt=50)       # MAX_NUM_GT_BOXES = 50
def step():
    opt.zero_grad(set_to_none=True)
    out = m(*ins)
    loss = out[3].mean() + out[4].mean() + out[5].mean() + out[6].mean() + out[7].mean()
    loss.backward(); opt.step()
    return out, loss
out, loss = step(); torch.cuda.synchronize()
print("rois", tuple(out[0].shape), "loss", float(loss), "finite grads", all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None))
step(); torch.cuda.synchronize(); t = time.time()
for _ in range(3): step()
torch.cuda.synchronize(); dt = (time.time() - t) / 3
print("cfg5-like step (bs=8, P=512, R101, COCO variant, %s AIT GEMMs): %.1f ms -> %.1f pairs/s; peak memory %.1f GB"
      % (ops.MATMUL_DTYPE, dt * 1e3, bs / dt, torch.cuda.max_memory_allocated() / 2**30))
