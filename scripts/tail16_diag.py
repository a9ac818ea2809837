"""diagnostic: the proposal tail node on bf16 storage against a float64 host emulation that rounds to bf16 at the same points"""
import sys
import torch
import torch.nn.functional as F
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ait_amd.faster_rcnn as fr
from ait_amd import ops


def r(t):
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


def emulate(m, dp, x, q):
    def sk(prefix, blk, x):
        fs = []
        for i in range(2):
            conv = blk.convs[i][0]
            w, b = dp["%s.convs.%d.0.weight" % (prefix, i)], dp["%s.convs.%d.0.bias" % (prefix, i)]
            rs = (lambda t: t) if "--sk-exact" in sys.argv else r
            fs.append(F.relu(F.conv2d(rs(x), rs(w), b, 2, conv.padding, 1, 8)))
        return r(fs[0] ** 2 + fs[1] ** 2)
    xt = torch.cat([sk("sk.sk_props", m.sk.sk_props, x), sk("sk.sk_query", m.sk.sk_query, q)])
    for k, b in enumerate(m.RCNN_top[0]):
        def fold(name, bn):
            scale, shift, _ = fr._bn_affine(bn)
            w = dp["RCNN_base.backbone.layer4.%d.%s.weight" % (k, name)]
            return r(w * scale.cpu().double()[:, None, None, None]), shift.cpu().double()[None, :, None, None]
        w1, s1 = fold("conv1", b.bn1)
        a1 = r(F.relu(F.conv2d(xt, w1) + s1))
        w2, s2 = fold("conv2", b.bn2)
        a2 = r(F.relu(F.conv2d(a1, w2, padding=1) + s2))
        idn = xt
        if k == 0:
            wd, sd = fold("downsample.0", b.downsample[1])
            idn = r(F.conv2d(xt, wd) + sd)
        w3, s3 = fold("conv3", b.bn3)
        xt = r(F.relu(F.conv2d(a2, w3) + s3 + idn))
    return xt.mean((2, 3))


if __name__ == "__main__":
    torch.manual_seed(7)
    m = fr.resnet(('__background__', 'fg'), 50, pretrained=False, class_agnostic=True, num_K=3)
    m.create_architecture()
    for mod in m.RCNN_top.modules():
        if isinstance(mod, torch.nn.BatchNorm2d):
            mod.running_mean.normal_(0, 0.1); mod.running_var.uniform_(0.5, 1.5); mod.weight.data.uniform_(0.5, 1.5); mod.bias.data.normal_(0, 0.1)
    if "--positive" in sys.argv:                       # every pre-activation positive: no ReLU ever masks, no mask can flip
        for mod in m.RCNN_top.modules():
            if isinstance(mod, torch.nn.BatchNorm2d):
                mod.weight.data.mul_(0.002)
                mod.running_mean.zero_()
                mod.bias.data.fill_(1.0)
        for blk in (m.sk.sk_props, m.sk.sk_query):
            for c in blk.convs:
                c[0].bias.data.fill_(8.0)
    m = m.cuda().train()
    names = [n for n, _ in m.named_parameters()
             if n.startswith(("sk.sk_props.convs", "sk.sk_query.convs", "RCNN_base.backbone.layer4.", "RCNN_top."))
             and "bn" not in n and "downsample.1" not in n]
    rel = lambda a, b: float((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30))
    for bp, bs in ((37, 2), (100, 4)):
        x0 = torch.randn(bp, 1024, 8, 8, device="cuda").contiguous(memory_format=torch.channels_last)
        q0 = torch.randn(bs, 1024, 8, 8, device="cuda")
        cot = torch.randn(bp + bs, 2048, device="cuda")
        m.zero_grad(set_to_none=True)
        x, q = x0.clone().requires_grad_(True), q0.clone().requires_grad_(True)
        ops.set_matmul_dtype("bf16")
        try:
            yp, yq = m._tail(x, q)
            (torch.cat([yp, yq]) * cot).sum().backward()
        finally:
            ops.set_matmul_dtype("f32")
        params = dict(m.named_parameters())
        got = [torch.cat([yp, yq]).detach(), x.grad.clone(), q.grad.clone()] + [params[n].grad.clone() for n in names]
        dp = {n: params[n].detach().cpu().double().requires_grad_(True) for n in names}
        xe, qe = x0.cpu().double().requires_grad_(True), q0.cpu().double().requires_grad_(True)
        ye = emulate(m, dp, xe, qe)
        (ye * cot.cpu().double()).sum().backward()
        want = [ye.detach(), xe.grad, qe.grad] + [dp[n].grad for n in names]
        print("bp, bs =", bp, bs)
        for lab, a, b in zip(["pooled", "d_x_props", "d_x_query"] + names, got, want):
            print("  %-55s rel %.5f" % (lab, rel(a, b)))
        # structure of the error of the last bottleneck's conv3 gradient
        n3 = "RCNN_base.backbone.layer4.2.conv3.weight"
        a = params[n3].grad.double().cpu().view(2048, 512)
        b = dp[n3].grad.view(2048, 512)
        e = a - b
        print("   conv3.2: per-row rel min/med/max", [round(float(v), 5) for v in torch.quantile(e.norm(dim=1) / b.norm(dim=1), torch.tensor([0., .5, 1.], dtype=torch.double))])
        print("   conv3.2: per-col rel min/med/max", [round(float(v), 5) for v in torch.quantile(e.norm(dim=0) / b.norm(dim=0), torch.tensor([0., .5, 1.], dtype=torch.double))])
        print("   conv3.2: best scalar fit a ~ c*b: c =", float((a * b).sum() / (b * b).sum()), " residual rel", float((a - b * ((a * b).sum() / (b * b).sum())).norm() / b.norm()))
        a, b = got[0].double().cpu(), want[0]
        per_map = (a - b).norm(dim=1) / b.norm(dim=1)
        print("   pooled per map: min %.2e med %.2e max %.2e; last 6:" % (float(per_map.min()), float(per_map.median()), float(per_map.max())), [round(float(v), 5) for v in per_map[-6:]])
        per_ch = (a - b).norm(dim=0) / (b.norm(dim=0) + 1e-30)
        print("   pooled per channel: min %.2e med %.2e max %.2e" % (float(per_ch.min()), float(per_ch.median()), float(per_ch.max())))
        d = (a - b).abs() / (b.abs() + 1e-6)
        print("   pooled elementwise rel: median %.2e, 90%% %.2e, 99%% %.2e" % tuple(float(v) for v in torch.quantile(d.flatten()[:1000000], torch.tensor([.5, .9, .99], dtype=torch.double))))
