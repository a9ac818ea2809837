"""The reference's training loop (trainval_net_voc.py:277-305, 335-423) driven through the drop-in
imports of INTEGRATION.md section 3, on synthetic (target, query) pairs.

    python examples/train_synthetic.py --bs 4 --proposals 128 --steps 5

The only lines that differ from the reference driver are the imports and the data source.
"""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ait_amd.config import cfg, cfg_from_list          # reference: model.utils.config  # noqa: E402
from ait_amd.faster_rcnn import resnet                  # reference: model.faster_rcnn.resnet_sys_transformer_sk_dilat  # noqa: E402
import bench                                           # synthetic inputs only  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bs", type=int, default=2)
    ap.add_argument("--proposals", type=int, default=128)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--lr", type=float, default=0.001)
    args = ap.parse_args()

    cfg_from_list(['TRAIN.BATCH_SIZE', args.proposals])
    np.random.seed(cfg.RNG_SEED)
    fasterRCNN = resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    fasterRCNN.create_architecture()

    lr, params = args.lr, []
    for key, value in dict(fasterRCNN.named_parameters()).items():     # trainval_net_voc.py:289-296
        if value.requires_grad:
            if 'bias' in key:
                params += [{'params': [value], 'lr': lr, 'weight_decay': 0}]
            else:
                params += [{'params': [value], 'lr': lr, 'weight_decay': 0.0001}]
    fasterRCNN.cuda()
    optimizer = torch.optim.SGD(params, momentum=0.9)
    fasterRCNN.train()
    for step in range(args.steps):
        im_data, query, im_info, gt_boxes, num_boxes = bench.synth_batch(args.bs, 100 + step, "cuda")
        fasterRCNN.zero_grad()
        rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_box, RCNN_loss_cls, margin_loss, \
            RCNN_loss_box, rois_label, _ = fasterRCNN(im_data, query, im_info, gt_boxes, num_boxes)
        cost = rpn_loss_cls.mean() + rpn_loss_box.mean() + RCNN_loss_cls.mean() \
            + RCNN_loss_box.mean() + margin_loss.mean()
        optimizer.zero_grad()
        cost.backward()
        optimizer.step()
        fg_cnt = int(torch.sum(rois_label.data.ne(0)))
        print("[step %d] loss %.4f rpn_cls %.4f rpn_box %.4f rcnn_cls %.4f rcnn_box %.4f margin %.4f fg/bg %d/%d"
              % (step, cost.item(), rpn_loss_cls.item(), rpn_loss_box.item(), RCNN_loss_cls.item(),
                 RCNN_loss_box.item(), margin_loss.item(), fg_cnt, rois_label.numel() - fg_cnt))
    torch.save({'session': 1, 'epoch': 1, 'model': fasterRCNN.state_dict(), 'optimizer': optimizer.state_dict(),
                'pooling_mode': cfg.POOLING_MODE, 'class_agnostic': True}, "/tmp/ait_amd_example_ckpt.pth")
    print("checkpoint with the reference's layout written to /tmp/ait_amd_example_ckpt.pth")


if __name__ == "__main__":
    main()
