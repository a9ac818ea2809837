"""The SK block -> layer4 shortcut of ait_amd.faster_rcnn._fasterRCNN.forward: layer4 opens with
stride-2 1x1 convolutions (resnet_sys_transformer_sk_dilat.py:78, 482-490), so only the even
positions of the SK block's output are read; evaluating SK at stride 2 and layer4's first block at
stride 1 must give the same pooled features AND the same gradients (float64, CPU, exact up to
summation order)."""
import torch


def test_stride2_sk_equals_full_sk_through_layer4():
    from ait_amd.faster_rcnn import resnet
    torch.manual_seed(0)
    m = resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    m = m.double().eval()
    assert m._top_stride() == 2
    x = torch.randn(3, 1024, 8, 8, dtype=torch.float64, requires_grad=True)
    cot = torch.randn(3, 2048, dtype=torch.float64)
    params = [p for p in list(m.sk.sk_props.convs.parameters()) + list(m.RCNN_top.parameters()) if p.requires_grad]

    def run(stride):
        y = m._head_to_tail(m.sk.sk_props(x, stride), subsampled=stride != 1)
        g = torch.autograd.grad(y, [x] + params, cot, allow_unused=True)
        return y, g

    y1, g1 = run(1)
    y2, g2 = run(2)
    assert (m.sk.sk_props(x, 2) - m.sk.sk_props(x)[:, :, ::2, ::2]).abs().max() < 1e-12
    assert (y1 - y2).abs().max() < 1e-12 * y1.abs().max()
    for a, b in zip(g1, g2):
        assert (a is None) == (b is None)
        if a is not None:
            assert (a - b).abs().max() <= 1e-11 * max(1.0, float(a.abs().max()))


def test_backbone_stage_boundaries_skip_dead_positions_exactly():
    """run_stages: the last block of layer1 / layer2 computes only what layer2[0] / layer3[0] read."""
    from ait_amd import faster_rcnn as fr
    torch.manual_seed(1)
    net = fr.resnet50().double().eval()
    for p in net.parameters():
        p.requires_grad_(True)
    x = torch.randn(2, 64, 19, 31, dtype=torch.float64, requires_grad=True)      # odd sizes on purpose
    stages = [net.layer1, net.layer2, net.layer3]
    params = [p for st in stages for p in st.parameters() if p.dim() == 4]
    y_ref = net.layer3(net.layer2(net.layer1(x)))
    cot = torch.randn_like(y_ref)
    g_ref = torch.autograd.grad(y_ref, [x] + params, cot)
    y = fr.run_stages(stages, x)
    g = torch.autograd.grad(y, [x] + params, cot)
    assert y.shape == y_ref.shape and (y - y_ref).abs().max() < 1e-12 * y_ref.abs().max()
    for a, b in zip(g, g_ref):
        assert (a - b).abs().max() <= 1e-11 * max(1.0, float(b.abs().max()))
