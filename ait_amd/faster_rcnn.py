"""Detector assembly around the AIT hot path: host-side mirror of

  lib/model/faster_rcnn/faster_rcnn_sys_transformer_sk_dilat.py   (_fasterRCNN, CoAttentionModule)
  lib/model/faster_rcnn/resnet_sys_transformer_sk_dilat.py        (ResNet, RCNNBackbone, resnet)
  lib/model/modules/blocks_sys_transformer_sk_dilat.py:915-997    (SKBlock, SKNet)

with the reference's constructor/forward signatures, 10-tuple return value and state_dict key
names (a reference checkpoint loads with load_state_dict).  The hot path inside -- RoIAlign,
the AIT Transformer, NMS -- runs in libait_hip.so; the ResNet convolutions stay on
PyTorch-ROCm (MIOpen), as SURVEY.md section 2 scopes them.

    model = resnet(classes, 50, pretrained=False, class_agnostic=True, num_K=3)
    model.create_architecture()
    rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, margin_loss, \
        RCNN_loss_bbox, rois_label, c_att = model(image, query, img_info, gt_boxes, num_boxes)
"""
import math
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .config import cfg
from .roi_layers import ROIAlign
from .rpn import _ProposalTargetLayer, _RPN, _smooth_l1_loss
from .system import MultiHeadAttention, Transformer, _Linear, _split_k, conv2d_1x1


# ------------------------------------------------------------------------------------------
# channel block between AIT and layer4
# ------------------------------------------------------------------------------------------
_SK_FULL = os.environ.get("AIT_SK_FULL", "0") == "1"
# Proposal tail (AIT output -> SK block -> layer4) in channels-last memory: the AIT's token-major
# GEMM output IS channels-last, and MIOpen's fastest fp32 kernels for these shapes are its NHWC
# implicit-GEMM ones, which on NCHW tensors pay a layout transpose in and out of every call
# (measured: 86.8 -> 82.5 ms/step).  AIT_TOP_NHWC=0 keeps NCHW.
_TOP_NHWC = os.environ.get("AIT_TOP_NHWC", "1") == "1"
# The C4 trunk likewise (82.5 -> 80.5 ms/step); its output is handed on in NCHW.  AIT_BASE_NHWC=0 keeps NCHW.
_BASE_NHWC = os.environ.get("AIT_BASE_NHWC", "1") == "1"
# RoIAlign on channels-last features writing the token rows the AIT embedding reads (no NCHW <->
# token transposes of the 235 MB pooled tensor, coalesced C-vector taps).  AIT_ROI_NHWC=0: NCHW.
_ROI_NHWC = os.environ.get("AIT_ROI_NHWC", "1") == "1"


def _fmt(x):
    cl = x.dim() == 4 and not x.is_contiguous() and x.is_contiguous(memory_format=torch.channels_last)
    return torch.channels_last if cl else torch.contiguous_format


class _SkSqSum(torch.autograd.Function):
    """relu(a)^2 + relu(b)^2 (the SKBlock tail as the reference executes it)."""

    @staticmethod
    def forward(ctx, a, b):
        ctx.save_for_backward(a, b)
        return ops.sk_sqsum_fwd(a, b)

    @staticmethod
    def backward(ctx, dy):
        a, b = ctx.saved_tensors
        return ops.sk_sqsum_bwd(dy.contiguous(memory_format=_fmt(a)), a, b)


# SK's two grouped convolutions (8 groups of 128 channels, 1x1 and 3x3) on the library's implicit-GEMM kernels:
# with 128 channels per group a 128-column tile of the GEMM lies inside one group, so the grouped convolution is
# the dense kernel with the gathered operand's channel offset taken from the tile's column (csrc/gemm_f32_impl.h,
# ConvGeom::a_group).  Proposal side only (bs*64 query rows are a launch-latency problem: MIOpen).  Opt-in
# (AIT_SK_HIP=1): built and tested, but 2.6 ms/step SLOWER than MIOpen / CK on the bench -- the block runs at
# stride 2 (dead-position elimination), and the data gradient of a stride-2 convolution as a gather over all nine
# taps multiplies three zero rows for every useful one (181 GFLOP executed for 45 useful).
_SK_HIP = os.environ.get("AIT_SK_HIP", "0") == "1"
_SK_HIP_MIN_ROWS = 4096


class _GroupedConv(torch.autograd.Function):
    """y = conv2d(x, w, bias, stride, padding, groups) on channels-last maps, forward and both gradients."""

    @staticmethod
    def forward(ctx, x, w, bias, stride, pad, groups):
        n, c, h, wd = x.shape
        k = w.size(2)
        oh, ow = (h + 2 * pad - k) // stride + 1, (wd + 2 * pad - k) // stride + 1
        xr = x.permute(0, 2, 3, 1).reshape(n * h * wd, c)                 # (channels-last: a view)
        wr = w.permute(0, 2, 3, 1).contiguous()                           # [cout, kh, kw, cin/g]
        geom = ops.conv_geom(n, (h, wd), (oh, ow), (k, k), stride, pad, groups)
        y = ops.conv_fwd(xr, wr, geom, bias=bias)
        ctx.save_for_backward(xr, wr)
        ctx.cfg = (geom, k, (n, c, h, wd), (oh, ow), bias is not None)
        return y.view(n, oh, ow, w.size(0)).permute(0, 3, 1, 2)           # NCHW shape, channels-last memory

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        geom, k, (n, c, h, wd), (oh, ow), has_bias = ctx.cfg
        dyr = dy.permute(0, 2, 3, 1).reshape(n * oh * ow, dy.size(1))
        if not dyr.is_contiguous():
            dyr = dyr.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = ops.conv_bwd_data(dyr, wr, geom).view(n, h, wd, c).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = ops.conv_bwd_weight(dyr, xr, geom, k, k, split_k=16).permute(0, 3, 1, 2)
        if has_bias and ctx.needs_input_grad[2]:
            db = ops.colsum(dyr)
        return dx, dw, db, None, None, None


class SKBlock(nn.Module):
    """Selective-kernel block as the reference actually computes it: two grouped conv branches
    (1x1 and 3x3, 8 groups, ReLU); the branch-attention weights `a` are computed and then NOT
    used -- the output is sum_branches f*f (blocks_sys_transformer_sk_dilat.py:974-981)."""

    def __init__(self, channels, reduction=16):
        super().__init__()
        kernels = [1, 3]
        self.n_state = len(kernels)
        self.convs = nn.ModuleList([nn.Sequential(
            nn.Conv2d(channels, channels, kernel_size=k, stride=1, padding=k // 2, groups=8),
            nn.ReLU(inplace=True)) for k in kernels])
        self.gap = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(channels, channels // reduction)
        self.sk = nn.Linear(channels // reduction, channels * self.n_state)
        self.softmax = nn.Softmax(dim=1)
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Conv2d)):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
                nn.init.constant_(m.bias, 0)

    def _branch(self, i, x, stride):
        conv = self.convs[i][0]
        k, g = conv.kernel_size[0], conv.groups
        if (_SK_HIP and x.is_cuda and x.dtype == torch.float32 and _fmt(x) == torch.channels_last
                and x.size(0) * x.size(2) * x.size(3) >= _SK_HIP_MIN_ROWS and conv.dilation == (1, 1)
                and (x.size(1) // g) % 128 == 0 and (conv.out_channels // g) % 128 == 0):
            oh = (x.size(2) + 2 * conv.padding[0] - k) // stride + 1
            ow = (x.size(3) + 2 * conv.padding[1] - k) // stride + 1
            if ops.conv_supported((x.size(2), x.size(3)), (oh, ow), stride, x.size(1) // g, conv.out_channels // g):
                return _GroupedConv.apply(x, conv.weight, conv.bias, stride, conv.padding[0], g)
        if stride == 1:
            return conv(x)
        return F.conv2d(x, conv.weight, conv.bias, stride, conv.padding, conv.dilation, conv.groups)

    def forward(self, x, stride=1):
        """stride = 2 evaluates the block only at the even output positions (see
        _fasterRCNN.forward: the only consumer, layer4's stride-2 1x1 convolutions, never reads the
        others); the values at those positions are the same convolution sums."""
        if x.is_cuda and x.dtype == torch.float32 and self.n_state == 2 and x.numel() % 4 == 0:
            # convolutions on MIOpen, then ReLU / square / branch sum in one fused HIP pass
            a = self._branch(0, x, stride)
            fmt = _fmt(a)
            return _SkSqSum.apply(a.contiguous(memory_format=fmt),
                                  self._branch(1, x, stride).contiguous(memory_format=fmt))
        out = None
        for i in range(self.n_state):
            f = F.relu(self._branch(i, x, stride))
            out = f * f if out is None else out + f * f
        return out


class SKNet(nn.Module):
    def __init__(self, channels, reduction=16):
        super().__init__()
        self.sk_props = SKBlock(channels, reduction)
        self.sk_query = SKBlock(channels, reduction)

    def forward(self, x_props, x_query, stride=1):
        return self.sk_props(x_props, stride), self.sk_query(x_query, stride)


# ------------------------------------------------------------------------------------------
# image-level co-attention (VOC variant): two MultiHeadAttention blocks with len_q = H_i*W_i
# ------------------------------------------------------------------------------------------
class CoAttentionModule(nn.Module):
    def __init__(self, d_word_vec, d_model, d_inner, n_head, d_k, d_v, dropout=0.1):
        super().__init__()
        self.d_model, self.d_word_vec = d_model, d_word_vec
        self.img_emb = nn.Sequential(conv2d_1x1(d_word_vec, d_model, bias=True))
        self.qry_emb = nn.Sequential(conv2d_1x1(d_word_vec, d_model, bias=True))
        self.i2q_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.q2i_attn = MultiHeadAttention(n_head, d_model, d_k, d_v, dropout=dropout)
        self.img_trans = nn.Sequential(nn.Linear(d_model, d_word_vec, bias=True))
        self.qry_trans = nn.Sequential(nn.Linear(d_model, d_word_vec, bias=True))

    def forward(self, x_img, x_qry):
        bs, _, h_i, w_i = x_img.size()
        _, _, h_q, w_q = x_qry.size()
        if x_img.is_cuda and x_img.dtype == torch.float32 and _fmt(x_img) == torch.channels_last:
            # channels-last features ARE the token rows [bs*HW, C] the attention works on: the 1x1
            # embeddings run as token-major GEMMs and the result is handed on as a channels-last
            # view -- no NCHW <-> token transposes on either side (same arithmetic, same values)
            c = x_img.size(1)
            tok = x_img.permute(0, 2, 3, 1).reshape(bs * h_i * w_i, c)
            e = self.img_emb[0]
            img = _Linear.apply(tok, e.weight.view(e.out_channels, c), e.bias).view(bs, h_i * w_i, -1)
            qry = self.qry_emb(x_qry).flatten(2).transpose(1, 2)
            enc_img, _ = self.q2i_attn(q=img, k=qry, v=qry, mask=None)
            enc_qry, _ = self.i2q_attn(q=qry, k=img, v=img, mask=None)
            non_img = self.img_trans(enc_img).view(bs, h_i, w_i, self.d_word_vec).permute(0, 3, 1, 2)
            non_qry = self.qry_trans(enc_qry).transpose(1, 2).reshape(bs, self.d_word_vec, h_q, w_q)
            return non_img, non_qry
        img = self.img_emb(x_img).flatten(2).transpose(1, 2)          # [bs, HW, 512]
        qry = self.qry_emb(x_qry).flatten(2).transpose(1, 2)          # [bs, 64, 512]
        enc_img, _ = self.q2i_attn(q=img, k=qry, v=qry, mask=None)
        enc_qry, _ = self.i2q_attn(q=qry, k=img, v=img, mask=None)
        non_img = self.img_trans(enc_img).transpose(1, 2).reshape(bs, self.d_word_vec, h_i, w_i)
        non_qry = self.qry_trans(enc_qry).transpose(1, 2).reshape(bs, self.d_word_vec, h_q, w_q)
        return non_img, non_qry


class _Bmm(torch.autograd.Function):
    """Batched C = alpha * op(A) op(B) on the matrix cores (one launch per product, forward and backward)."""

    @staticmethod
    def forward(ctx, a, b, trans_a, trans_b, alpha):
        ctx.save_for_backward(a, b)
        ctx.cfg = (trans_a, trans_b, alpha)
        return ops.bgemm(a, b, trans_a, trans_b, alpha)

    @staticmethod
    def backward(ctx, dc):
        a, b = ctx.saved_tensors
        ta, tb, alpha = ctx.cfg
        da = db = None
        if not ta and tb:            # C = A B^T
            da = ops.bgemm(dc, b, False, False, alpha) if ctx.needs_input_grad[0] else None     # dC B
            db = ops.bgemm(dc, a, True, False, alpha) if ctx.needs_input_grad[1] else None      # dC^T A
        elif not ta and not tb:      # C = A B
            da = ops.bgemm(dc, b, False, True, alpha) if ctx.needs_input_grad[0] else None      # dC B^T
            db = ops.bgemm(a, dc, True, False, alpha) if ctx.needs_input_grad[1] else None      # A^T dC
        elif ta and not tb:          # C = A^T B
            da = ops.bgemm(b, dc, False, True, alpha) if ctx.needs_input_grad[0] else None      # B dC^T
            db = ops.bgemm(a, dc, False, False, alpha) if ctx.needs_input_grad[1] else None     # A dC
        else:
            raise NotImplementedError
        return da, db, None, None, None


class CoAttention(nn.Module):
    """Non-local image <-> query co-attention of the COCO variant
    (lib/model/modules/blocks_coatt_transformer_sk.py:17-122, 'division' normalisation,
    GroupNorm(32) on both output projections, zero-initialised GroupNorm affine)."""

    def __init__(self, **kwargs):
        super().__init__()
        self.in_ch = kwargs.get('in_ch', 1024)
        self.c_hidden = kwargs.get('c_hidden', 512)
        self.with_residual = kwargs.get('with_residual', True)
        self.normlization = kwargs.get('normlization', 'division')
        self.emb = conv2d_1x1(self.in_ch, self.c_hidden)
        self.rho = conv2d_1x1(self.in_ch, self.c_hidden)
        self.phi = conv2d_1x1(self.in_ch, self.c_hidden)
        self.omega = nn.Sequential(conv2d_1x1(self.c_hidden, self.in_ch), nn.GroupNorm(32, self.in_ch))
        self.theta = nn.Sequential(conv2d_1x1(self.c_hidden, self.in_ch), nn.GroupNorm(32, self.in_ch))
        if self.normlization == 'softmax':
            self.softmax = nn.Softmax(dim=2)
        for m in self.modules():
            if isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 0)
                nn.init.constant_(m.bias, 0)

    def _forward_hip(self, x_img, x_qry):
        """Token-major on the library's kernels: the five 1x1 embeddings as GEMMs over the channels-last
        token rows, the three non-local products as batched GEMMs (ait_gemm_f32_batched).  The 'division'
        normalisation is the products' alpha; GroupNorm and the residual stay on torch."""
        bz, C, h_i, w_i = x_img.shape
        _, _, h_q, w_q = x_qry.shape
        ch, n_i, n_q = self.c_hidden, h_i * w_i, h_q * w_q
        ti = x_img.permute(0, 2, 3, 1).reshape(bz * n_i, C)
        tq = x_qry.permute(0, 2, 3, 1).reshape(bz * n_q, C)
        lin = lambda conv, t: _Linear.apply(t, conv.weight.view(conv.out_channels, -1), conv.bias)
        emb_img = lin(self.emb, ti).view(bz, n_i, ch)
        emb_qry = lin(self.emb, tq).view(bz, n_q, ch)
        rho_qry = lin(self.rho, tq).view(bz, n_q, ch)
        phi_img = lin(self.phi, ti).view(bz, n_i, ch)                     # (tokens: the reference's [bz, ch, N_i] transposed)
        rel = _Bmm.apply(rho_qry, phi_img, False, True, 1.0)              # [bz, N_q, N_i]
        non_img_t = _Bmm.apply(rel, emb_qry, True, False, 1.0 / n_q)      # (rel^T / N_q) emb_qry  [bz, N_i, ch]
        non_qry_t = _Bmm.apply(rel, emb_img, False, False, 1.0 / n_i)     # (rel / N_i) emb_img    [bz, N_q, ch]
        om, th = self.omega[0], self.theta[0]
        non_img = lin(th, non_img_t.reshape(bz * n_i, ch)).view(bz, h_i, w_i, C).permute(0, 3, 1, 2)
        non_qry = lin(om, non_qry_t.reshape(bz * n_q, ch)).view(bz, h_q, w_q, C).permute(0, 3, 1, 2)
        non_img, non_qry = self.theta[1](non_img), self.omega[1](non_qry)
        if self.with_residual:
            non_img, non_qry = non_img + x_img, non_qry + x_qry
        return non_img, non_qry

    def forward(self, x_img, x_qry):
        if (x_img.is_cuda and x_img.dtype == torch.float32 and self.normlization == 'division'
                and os.environ.get("AIT_COATT_TORCH", "0") != "1"):
            return self._forward_hip(x_img, x_qry)
        bz, _, h_i, w_i = x_img.shape
        _, _, h_q, w_q = x_qry.shape
        ch = self.c_hidden
        emb_img = self.emb(x_img).view(bz, ch, -1).transpose(1, 2)        # [bz, N_i, ch]
        emb_qry = self.emb(x_qry).view(bz, ch, -1).transpose(1, 2)        # [bz, N_q, ch]
        rho_qry = self.rho(x_qry).view(bz, ch, -1).transpose(1, 2)        # [bz, N_q, ch]
        phi_img = self.phi(x_img).view(bz, ch, -1)                        # [bz, ch, N_i]
        rel = torch.matmul(rho_qry, phi_img)                              # [bz, N_q, N_i]
        n_q, n_i = rel.size(1), rel.size(2)
        q2i, i2q = rel, rel.transpose(1, 2)
        if self.normlization == 'softmax':
            q2i, i2q = self.softmax(q2i), self.softmax(i2q.contiguous())
        else:
            q2i, i2q = q2i / n_i, i2q / n_q
        non_img = self.theta(torch.matmul(i2q, emb_qry).transpose(1, 2).reshape(bz, ch, h_i, w_i))
        non_qry = self.omega(torch.matmul(q2i, emb_img).transpose(1, 2).reshape(bz, ch, h_q, w_q))
        if self.with_residual:
            non_img, non_qry = non_img + x_img, non_qry + x_qry
        return non_img, non_qry


class CoAttentionModuleCOCO(nn.Module):
    """faster_rcnn_coatt_transformer_sk.py:104-158: thin wrapper (keeps the checkpoint key
    prefix coattention_module.coattention.*)."""

    def __init__(self, inplanes):
        super().__init__()
        self.coattention = CoAttention(in_ch=inplanes, c_hidden=max(1, inplanes // 2),
                                       with_residual=True, normlization='division')

    def forward(self, x_img, x_qry):
        if x_img.is_cuda:           # token-major kernels: channels-last features are taken as they are
            return self.coattention(x_img, x_qry)
        return self.coattention(x_img.contiguous(), x_qry.contiguous())


# ------------------------------------------------------------------------------------------
# frozen BatchNorm (+ residual) (+ ReLU) as one HIP pass
# ------------------------------------------------------------------------------------------
class _Subsample(torch.autograd.Function):
    """x[:, :, ::s, ::s] in the memory format of x.  The stock slice backward materialises its
    zero-filled gradient in NCHW order, which then has to be re-laid-out before it can meet the
    channels-last gradient of the other consumer of x; this one keeps the format."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.shape, ctx.s, ctx.fmt = x.shape, s, _fmt(x)
        return x[:, :, ::s, ::s].contiguous(memory_format=ctx.fmt)

    @staticmethod
    def backward(ctx, dy):
        g = torch.empty(ctx.shape, dtype=dy.dtype, device=dy.device, memory_format=ctx.fmt).zero_()
        g[:, :, ::ctx.s, ::ctx.s] = dy
        return g, None


class _BnAct(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        y = ops.bn_act_fwd(x, scale, shift, residual, relu)
        ctx.save_for_backward(y if relu else None, scale)
        ctx.relu = relu
        ctx.has_res = residual is not None
        ctx.fmt = _fmt(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        dy = dy.contiguous(memory_format=ctx.fmt)
        dx, dres = ops.bn_act_bwd(dy, y, scale, ctx.relu, ctx.has_res and ctx.needs_input_grad[3])
        return dx, None, None, dres, None


def _bn_frozen(bn):
    return (not bn.training) and not bn.weight.requires_grad and not bn.bias.requires_grad


def _bn_affine(bn):
    """(scale, shift) of a frozen eval-mode BatchNorm2d: y = x*scale[c] + shift[c]; cached until a
    buffer or parameter of the module changes."""
    key = (bn.weight._version, bn.bias._version, bn.running_mean._version, bn.running_var._version,
           bn.weight.data_ptr(), bn.running_mean.data_ptr())
    cache = getattr(bn, "_ait_affine", None)
    if cache is None or cache[0] != key:
        with torch.no_grad():
            scale = (bn.weight / torch.sqrt(bn.running_var + bn.eps)).float().contiguous()
            shift = (bn.bias - bn.running_mean * scale).float().contiguous()
        cache = (key, scale, shift, torch.ones_like(scale))
        bn._ait_affine = cache
    return cache[1], cache[2], cache[3]


def bn_act(x, bn, residual=None, relu=True):
    """relu(bn(x) + residual) for a FROZEN BatchNorm2d in eval mode (the only state the reference
    ever runs its BatchNorms in); anything else goes through torch."""
    if not (_bn_frozen(bn) and x.is_cuda and x.dtype == torch.float32):
        y = bn(x)
        if residual is not None:
            y = y + residual
        return F.relu(y) if relu else y
    scale, shift, _ = _bn_affine(bn)
    fmt = _fmt(x)
    x = x.contiguous(memory_format=fmt)
    if residual is not None:
        residual = residual.contiguous(memory_format=fmt)
    return _BnAct.apply(x, scale, shift, residual, relu)


# ------------------------------------------------------------------------------------------
# 1x1 convolution + frozen BN (+ residual) (+ ReLU) on channels-last activations = ONE GEMM
# ------------------------------------------------------------------------------------------
class _Conv1x1BnAct(torch.autograd.Function):
    """A stride-1 1x1 convolution over a channels-last tensor is the token-major product
    [N*H*W, Cin] x [Cout, Cin]^T; the frozen BatchNorm's scale is folded into the weight rows and
    its shift, the residual and the ReLU ride in the GEMM epilogue (ait_gemm_f32).  Backward: one
    masking pass (dz = dy * [y > 0], which is also the residual's gradient), then the two products
    dx = dz W' and dW = scale * (dz^T x)."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, ones, residual, relu):
        n, cin, h, w = x.shape
        cout = weight.shape[0]
        xm = x.permute(0, 2, 3, 1).reshape(n * h * w, cin)                  # view of channels-last x
        w2 = weight.reshape(cout, cin) * scale[:, None]
        rm = None if residual is None else residual.permute(0, 2, 3, 1).reshape(n * h * w, cout)
        ym = ops.gemm(xm, w2, bias=shift, residual=rm, relu=relu, exact=True)
        y = ym.view(n, h, w, cout).permute(0, 3, 1, 2)
        ctx.save_for_backward(xm, w2, scale, ones, y if relu else None)
        ctx.relu, ctx.has_res, ctx.wshape = relu, residual is not None, weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        xm, w2, scale, ones, y = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        if ctx.relu:
            dz, _ = ops.bn_act_bwd(dy, y, ones, True, False)               # dy * [y > 0]
        else:
            dz = dy
        n, cout, h, w = dz.shape
        dzm = dz.permute(0, 2, 3, 1).reshape(n * h * w, cout)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            dx = ops.gemm(dzm, w2, trans_b=False, exact=True)
            dx = dx.view(n, h, w, -1).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            dw = ops.gemm(dzm, xm, trans_a=True, trans_b=False, exact=True,
                          split_k=_split_k(cout, xm.shape[1], xm.shape[0]))
            dw = (dw * scale[:, None]).view(ctx.wshape)
        dres = dz if ctx.has_res and ctx.needs_input_grad[5] else None
        return dx, dw, None, None, None, dres, None


class _Conv3x3BnAct(torch.autograd.Function):
    """k x k convolution (stride 1 or 2) + frozen BatchNorm (+ ReLU) over a channels-last tensor whose maps
    have power-of-two sides (the 4x4 maps of layer4 on the proposal tail): an implicit GEMM on the matrix-core
    kernel (ait_conv_fwd_f32), the BatchNorm's scale folded into the weights and its shift / the ReLU in the
    epilogue.  Backward: one masking pass dz = dy * [y > 0], then ait_conv_bwd_data_f32 and
    ait_conv_bwd_weight_f32 (dW = scale * (dz^T (*) x))."""

    @staticmethod
    def forward(ctx, x, weight, scale, shift, ones, relu, stride, pad):
        n, cin, h, w = x.shape
        cout, _, kh, kw = weight.shape
        oh, ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
        xm = x.permute(0, 2, 3, 1).reshape(n * h * w, cin)                  # view of channels-last x
        w2 = (weight.permute(0, 2, 3, 1) * scale.view(-1, 1, 1, 1)).contiguous()       # [cout, kh, kw, cin]
        geom = ops.conv_geom(n, (h, w), (oh, ow), (kh, kw), stride, pad)
        ym = ops.conv_fwd(xm, w2, geom, bias=shift, relu=relu)
        y = ym.view(n, oh, ow, cout).permute(0, 3, 1, 2)
        ctx.save_for_backward(xm, w2, scale, ones, y if relu else None)
        ctx.geom, ctx.relu, ctx.k, ctx.xshape = geom, relu, (kh, kw), (n, cin, h, w)
        return y

    @staticmethod
    def backward(ctx, dy):
        xm, w2, scale, ones, y = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dz = ops.bn_act_bwd(dy, y, ones, True, False)[0] if ctx.relu else dy
        n, cout = dz.shape[0], dz.shape[1]
        dzm = dz.permute(0, 2, 3, 1).reshape(-1, cout)
        dx = dw = None
        if ctx.needs_input_grad[0]:
            n_, cin, h, w = ctx.xshape
            dx = ops.conv_bwd_data(dzm, w2, ctx.geom).view(n_, h, w, cin).permute(0, 3, 1, 2)
        if ctx.needs_input_grad[1]:
            rows = dzm.shape[0]
            dw = ops.conv_bwd_weight(dzm, xm, ctx.geom, ctx.k[0], ctx.k[1], split_k=max(8, min(64, rows // 2048 // 8 * 8)))
            dw = (dw * scale.view(-1, 1, 1, 1)).permute(0, 3, 1, 2)         # [cout, cin, kh, kw] (channels-last strides)
        return dx, dw, None, None, None, None, None, None


# Proposal tail (RCNN_top = layer4 on the 4x4 maps behind the SK block) on the library's own matrix-core
# kernels: 1x1 convolutions as GEMMs, 3x3 convolutions as implicit GEMMs, the frozen BatchNorm / residual /
# ReLU in their epilogues.  OPT-IN (AIT_TOP_HIP=1): measured on MI355X at bs=4, P=300 the step
# takes 76.1 ms with it against 69.8 ms on MIOpen's NHWC implicit-GEMM assembly kernels -- the products have
# 19200 rows, i.e. 300 tiles of 256x128 on 512 workgroup slots, and run at 90-118 TFLOP/s against MIOpen's
# ~125; parity-tested either way (tests/test_gpu_ops.py).
_TOP_HIP = os.environ.get("AIT_TOP_HIP", "0") == "1"
_TOP_HIP_MIN_ROWS = 4096      # fewer rows (the query side: bs*16) are one workgroup's serial K loop: MIOpen


def conv3x3_bn_act(x, conv, bn, relu=True, stride=None):
    """relu(bn(conv(x))) for a bias-free k x k convolution; the implicit-GEMM path needs channels-last fp32
    GPU activations, a frozen BN and power-of-two map sides, everything else is conv (MIOpen) + bn_act."""
    stride = conv.stride[0] if stride is None else stride
    if (_TOP_HIP and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _bn_frozen(bn) and conv.groups == 1
            and conv.bias is None and conv.dilation == (1, 1) and conv.padding[0] == conv.padding[1]
            and x.is_contiguous(memory_format=torch.channels_last)):
        kh, kw = conv.kernel_size
        pad = conv.padding[0]
        h, w = x.shape[2], x.shape[3]
        oh, ow = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
        rows = x.shape[0] * oh * ow
        if ops.conv_supported((h, w), (oh, ow), stride, conv.in_channels, conv.out_channels) and rows % 16 == 0 \
                and rows >= _TOP_HIP_MIN_ROWS:
            scale, shift, ones = _bn_affine(bn)
            return _Conv3x3BnAct.apply(x, conv.weight, scale, shift, ones, relu, stride, pad)
    y = conv(x) if stride == conv.stride[0] else F.conv2d(x, conv.weight, None, stride, conv.padding)
    return bn_act(y, bn, relu=relu)


# OFF by default: measured on MI355X, MIOpen's NHWC implicit-GEMM assembly kernels beat
# ait_gemm_f32 on these shapes even with the BN/ReLU pass fused away (bs=4, P=300: 75.9 ms/step
# without, 78.2 with layer4 only, 81.2 with layer3+4, 82.9 with every 1x1).  Kept as an opt-in
# (AIT_CONV1X1_GEMM=1, AIT_CONV1X1_MIN_C=<min channels>) and as the parity-tested reference point
# for the next attempt (tests/test_gpu_ops.py).
_CONV1X1_GEMM = os.environ.get("AIT_CONV1X1_GEMM", "0") == "1"
_CONV1X1_MIN_C = int(os.environ.get("AIT_CONV1X1_MIN_C", "512"))


def conv1x1_bn_act(x, conv, bn, residual=None, relu=True, stride1=False, hip=False):
    """relu(bn(conv(x)) + residual) for a bias-free 1x1 convolution.  `stride1`: run the
    convolution at stride 1 whatever conv.stride says (the input is already subsampled).
    Channels-last fp32 GPU activations with a frozen BN take the single-GEMM path above; everything
    else is conv (MIOpen) + bn_act."""
    stride = (1, 1) if stride1 else conv.stride
    if hip and x.numel() // max(1, x.shape[1]) < _TOP_HIP_MIN_ROWS:
        hip = False
    if ((_CONV1X1_GEMM or hip) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and _bn_frozen(bn)
            and conv.kernel_size == (1, 1) and stride == (1, 1) and conv.groups == 1 and conv.bias is None
            and conv.padding == (0, 0) and conv.in_channels % 4 == 0 and conv.out_channels % 4 == 0
            and (hip or min(conv.in_channels, conv.out_channels) >= _CONV1X1_MIN_C)
            and x.is_contiguous(memory_format=torch.channels_last)
            and (residual is None or residual.is_contiguous(memory_format=torch.channels_last))):
        scale, shift, ones = _bn_affine(bn)
        return _Conv1x1BnAct.apply(x, conv.weight, scale, shift, ones, residual, relu)
    y = F.conv2d(x, conv.weight, None, stride) if stride1 else conv(x)
    return bn_act(y, bn, residual=residual, relu=relu)


# ------------------------------------------------------------------------------------------
# ResNet backbone (stride on the first 1x1 of a bottleneck, ceil-mode max-pool without padding)
# ------------------------------------------------------------------------------------------
class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, kernel_size=1, stride=stride, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, kernel_size=3, stride=1, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, kernel_size=1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def forward(self, x, subsampled=False, out_stride=1):
        """subsampled=True: `x` already holds only the positions this block's stride-s 1x1
        convolutions read (x[:, :, ::s, ::s]), so they run at stride 1.
        out_stride=s: the caller guarantees that this block's output is read ONLY by stride-s 1x1
        convolutions (the first block of the next stage); the block then produces just those
        positions -- conv2 runs at stride s (same 3x3 sums at the kept positions), conv3, the
        frozen BN, the residual and the ReLU are position-wise."""
        hip = getattr(self, "_ait_hip", False)            # set on the blocks of RCNN_top (see resnet._init_modules)
        out = conv1x1_bn_act(x, self.conv1, self.bn1, stride1=subsampled, hip=hip)
        if hip and out_stride == 1:
            out = conv3x3_bn_act(out, self.conv2, self.bn2)
        elif out_stride == 1:
            out = bn_act(self.conv2(out), self.bn2)
        else:
            out = bn_act(F.conv2d(out, self.conv2.weight, None, out_stride, self.conv2.padding), self.bn2)
        if self.downsample is None:
            identity = x if out_stride == 1 else _Subsample.apply(x, out_stride)
        elif out_stride != 1:
            raise ValueError("out_stride needs an identity shortcut")
        else:
            identity = conv1x1_bn_act(x, self.downsample[0], self.downsample[1], relu=False, stride1=subsampled, hip=hip)
        return conv1x1_bn_act(out, self.conv3, self.bn3, residual=identity, hip=hip)


def _c4_size(h, w):
    """feature size of the C4 trunk for an h x w image: 7x7 stride-2 pad-3 stem, 3x3 stride-2 ceil-mode
    max-pool without padding, two stride-2 stages (resnet_sys_transformer_sk_dilat.py:117-125,78)"""
    def one(n):
        n = (n + 6 - 7) // 2 + 1
        n = -((n - 3) // -2) + 1
        n = (n - 1) // 2 + 1
        return (n - 1) // 2 + 1
    return one(h), one(w)


def _opens_with_stride2_1x1(stage):
    b0 = stage[0]
    return isinstance(b0, Bottleneck) and b0.conv1.kernel_size == (1, 1) and b0.downsample is not None \
        and b0.downsample[0].kernel_size == (1, 1) and b0.downsample[0].stride == b0.conv1.stride \
        and b0.conv1.stride == (2, 2)


def run_stages(stages, x):
    """layer_k(...layer_1(x)) for consecutive ResNet stages.  Where stage k+1 opens with stride-2
    1x1 convolutions (Bottleneck.conv1 / downsample, resnet_sys_transformer_sk_dilat.py:78), three
    quarters of stage k's last block output is never read: that block then computes only the
    positions that are (see Bottleneck.forward).  Same values and gradients as the plain
    composition; AIT_SK_FULL=1 restores it."""
    subsampled = False
    for k, stage in enumerate(stages):
        nxt = stages[k + 1] if k + 1 < len(stages) else None
        skip = (not _SK_FULL) and nxt is not None and _opens_with_stride2_1x1(nxt) \
            and isinstance(stage[-1], Bottleneck) and stage[-1].downsample is None and len(stage) > 1
        for i, blk in enumerate(stage):
            x = blk(x, subsampled=(subsampled and i == 0), out_stride=2 if (skip and i == len(stage) - 1) else 1)
        subsampled = skip
    return x


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=0, ceil_mode=True)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AvgPool2d(7)
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2. / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = nn.Sequential(
                nn.Conv2d(self.inplanes, planes * block.expansion, kernel_size=1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        layers += [block(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)


def resnet50():
    return ResNet(Bottleneck, [3, 4, 6, 3])


def resnet101():
    return ResNet(Bottleneck, [3, 4, 23, 3])


class RCNNBackbone(nn.Module):
    """C4 trunk shared by the target image and the query patch (siamese).  Keeps the whole
    ResNet as `.backbone` (so the checkpoint keys RCNN_base.backbone.* exist, including the never
    used fc) and aliases stem / layer1-3 exactly like the reference."""

    def __init__(self, cfg_, backbone, **kwargs):
        super().__init__()
        self.backbone = backbone
        self.channels = kwargs.get('channels', 2048)
        self.with_contextual_relation = kwargs.get('with_contextual_relation', False)
        if self.with_contextual_relation:
            raise NotImplementedError("the contextual-relation (GRU) branch is dead in the reference")
        self.stem = nn.Sequential(backbone.conv1, backbone.bn1, backbone.relu, backbone.maxpool)
        for p in self.stem[0].parameters():
            p.requires_grad = False
        for p in self.stem[1].parameters():
            p.requires_grad = False
        self.layer1 = backbone.layer1
        self.layer2 = backbone.layer2
        self.layer3 = backbone.layer3

    def forward(self, x):
        if _BASE_NHWC and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        x = self.stem[3](bn_act(self.stem[0](x), self.stem[1]))        # conv1, bn1+relu, maxpool
        # (a channels-last trunk hands its feature on channels-last: the co-attention reads token rows,
        # the RPN convolutions and the RoIAlign used here take either format)
        return run_stages([self.layer1, self.layer2, self.layer3], x), None


# ------------------------------------------------------------------------------------------
# the detector
# ------------------------------------------------------------------------------------------
class _fasterRCNN(nn.Module):
    # 'voc' = faster_rcnn_sys_transformer_sk_dilat.py (MultiHeadAttention co-attention),
    # 'coco' = faster_rcnn_coatt_transformer_sk.py (non-local co-attention); everything after the
    # co-attention is identical in the two reference files
    variant = 'voc'

    def __init__(self, classes, class_agnostic, num_K):
        super().__init__()
        self.classes = classes
        self.n_classes = len(classes)
        self.class_agnostic = class_agnostic
        self.channels = self.dout_base_model
        self.num_K = num_K
        C = self.channels
        if self.variant == 'coco':
            self.coattention_module = CoAttentionModuleCOCO(C)
        else:
            self.coattention = CoAttentionModule(d_k=64, d_v=64, d_word_vec=C, d_model=C // 2,
                                                 d_inner=C * 2, n_head=8, dropout=0.1)
        self.RCNN_rpn = _RPN(C)
        self.RCNN_proposal_target = _ProposalTargetLayer(self.n_classes)
        if cfg.POOLING_MODE != 'align':
            raise NotImplementedError("only POOLING_MODE 'align' is built (every shipped yml uses it)")
        self.RCNN_roi_align = ROIAlign((cfg.POOLING_SIZE, cfg.POOLING_SIZE), 1.0 / 16.0, 0,
                                       channels_last=_ROI_NHWC)
        self.sk = SKNet(channels=C)
        self.transformer = Transformer(d_k=64, d_v=64, d_model=C // 2, d_word_vec=C // 2,
                                       d_inner=C * 2, n_position=8 * 8, n_layers=1, n_head=8,
                                       dropout=0.1)
        self.transformer.channels_last_out = _TOP_NHWC      # (only read on the GPU path)
        self.triplet_loss = torch.nn.MarginRankingLoss(margin=cfg.TRAIN.MARGIN)

    def forward(self, image, query, img_info, gt_boxes, num_boxes):
        bs = image.size(0)
        img_info, gt_boxes, num_boxes = img_info.data, gt_boxes.data, num_boxes.data
        if self.training and image.is_cuda:
            # the RPN's anchor targets depend only on the inputs: their device part is enqueued FIRST and the
            # two counts per image the host-side sampling needs cross PCIe while the backbone runs
            # (im_hw_hint: the loader writes the padded tensor's size into im_info, roibatchLoader.py:228-255; the
            # layer verifies the prediction on the device and redoes itself if it was wrong)
            self.RCNN_rpn.RPN_anchor_target.begin(gt_boxes, img_info, *_c4_size(image.size(2), image.size(3)),
                                                  im_hw_hint=(image.size(2), image.size(3)))

        image_feat, _ = self.RCNN_base(image)                 # [bs, 1024, H_i, W_i]
        query_feat, _ = self.RCNN_base(query)                 # [bs, 1024, 8, 8]
        if self.variant == 'coco':
            non_img, non_qry = self.coattention_module(image_feat, query_feat)
        else:
            non_img, non_qry = self.coattention(x_img=image_feat, x_qry=query_feat)

        rois, rpn_loss_cls, rpn_loss_bbox = self.RCNN_rpn(non_img, img_info, gt_boxes, num_boxes)
        if self.training:
            rois, rois_label, rois_target, rois_inside_ws, rois_outside_ws = \
                self.RCNN_proposal_target(rois, gt_boxes, num_boxes)
            rois_label = rois_label.view(-1).long()
            rois_target = rois_target.view(-1, rois_target.size(2))
            rois_inside_ws = rois_inside_ws.view(-1, rois_inside_ws.size(2))
            rois_outside_ws = rois_outside_ws.view(-1, rois_outside_ws.size(2))
        else:
            rois_label = None
            rpn_loss_cls = margin_loss = rpn_loss_bbox = 0
        num_props = rois.size(1)

        props_feat = self.RCNN_roi_align(non_img, rois.view(-1, 5))          # [bs*P, 1024, 7, 7]
        props_feat = self.transformer(x_props=props_feat, x_query=non_qry)   # [bs*P, 1024, 8, 8]
        # layer4 opens with stride-2 1x1 convolutions (Bottleneck.conv1 / downsample,
        # resnet_sys_transformer_sk_dilat.py:78,482-490): of the SK block's 8x8 output only the 16
        # even positions are ever read.  SK is position-wise after its convolutions, so it is
        # evaluated at those positions only (stride 2) and layer4 takes the result at stride 1:
        # same sums, same gradients (the dead positions receive exactly zero gradient in the
        # reference), 3/4 of the SK work not done.  AIT_SK_FULL=1 keeps the dead positions.
        sk_stride = 1 if _SK_FULL else self._top_stride()
        props_feat, query_feat = self.sk(x_props=props_feat, x_query=non_qry, stride=sk_stride)
        c_att = None
        props_feat = self._head_to_tail(props_feat, subsampled=sk_stride != 1)   # [bs*P, 2048]
        query_feat = self._head_to_tail(query_feat, subsampled=sk_stride != 1)   # [bs, 2048]

        bbox_pred = self.RCNN_bbox_pred(props_feat)
        stack_feat = torch.cat((props_feat.view(bs, num_props, -1),
                                query_feat.unsqueeze(1).expand(-1, num_props, -1)), dim=2).reshape(-1, 4096)
        score = self.RCNN_cls_score(stack_feat)                              # similarity logits
        score_prob = F.softmax(score, 1)[:, 1]

        RCNN_loss_cls = 0
        RCNN_loss_bbox = 0
        if self.training:
            score_label = rois_label.view(bs, -1).float()
            gt_map = (score_label.unsqueeze(1) - score_label.unsqueeze(-1)).abs()
            sp = score_prob.view(bs, -1)
            pr_map = (sp.unsqueeze(1) - sp.unsqueeze(-1)).abs()
            target = -((gt_map - 1) ** 2) + gt_map
            RCNN_loss_cls = F.cross_entropy(score, rois_label)
            margin_loss = 3 * self.triplet_loss(pr_map, gt_map, target)
            RCNN_loss_bbox = _smooth_l1_loss(bbox_pred, rois_target, rois_inside_ws, rois_outside_ws)

        cls_prob = score_prob.view(bs, num_props, -1)
        bbox_pred = bbox_pred.view(bs, num_props, -1)
        return rois, cls_prob, bbox_pred, rpn_loss_cls, rpn_loss_bbox, RCNN_loss_cls, margin_loss, \
            RCNN_loss_bbox, rois_label, c_att

    def _init_weights(self):
        def normal_init(m, mean, stddev, truncated=False):
            if truncated:
                m.weight.data.normal_().fmod_(2).mul_(stddev).add_(mean)
            else:
                m.weight.data.normal_(mean, stddev)
                m.bias.data.zero_()
        t = cfg.TRAIN.TRUNCATED
        normal_init(self.RCNN_rpn.RPN_Conv, 0, 0.01, t)
        normal_init(self.RCNN_rpn.RPN_cls_score, 0, 0.01, t)
        normal_init(self.RCNN_rpn.RPN_bbox_pred, 0, 0.01, t)
        normal_init(self.RCNN_cls_score[0], 0, 0.01, t)
        normal_init(self.RCNN_cls_score[1], 0, 0.01, t)
        normal_init(self.RCNN_bbox_pred, 0, 0.001, t)

    def create_architecture(self):
        self._init_modules()
        self._init_weights()
        # convolution weights live in the memory format their activations use, so that MIOpen's
        # NHWC kernels need no per-call weight re-layout (values, shapes, state_dict unchanged)
        if _BASE_NHWC:
            self.RCNN_base.to(memory_format=torch.channels_last)
        if _TOP_NHWC:
            self.RCNN_top.to(memory_format=torch.channels_last)
            self.sk.to(memory_format=torch.channels_last)


class resnet(_fasterRCNN):
    def __init__(self, classes, num_layers=101, pretrained=False, class_agnostic=False, num_K=3):
        self.dout_base_model = 1024
        self.pretrained = pretrained
        self.class_agnostic = class_agnostic
        self.num_layers = num_layers
        _fasterRCNN.__init__(self, classes, class_agnostic, num_K)

    def _init_modules(self):
        net = resnet50() if self.num_layers == 50 else resnet101()
        if self.pretrained:
            raise NotImplementedError("ImageNet weights are loaded with load_state_dict by the driver")
        self.RCNN_base = RCNNBackbone(cfg, backbone=net)
        self.RCNN_top = nn.Sequential(net.layer4)
        for blk in net.layer4:
            blk._ait_hip = _TOP_HIP
        self.RCNN_cls_score = nn.Sequential(nn.Linear(2048 * 2, 8), nn.Linear(8, 2))
        self.RCNN_bbox_pred = nn.Linear(2048, 4 if self.class_agnostic else 4 * self.n_classes)

        def set_bn_fix(m):
            if m.__class__.__name__.find('BatchNorm') != -1:
                for p in m.parameters():
                    p.requires_grad = False
        self.RCNN_base.apply(set_bn_fix)
        self.RCNN_top.apply(set_bn_fix)

    def train(self, mode=True):
        nn.Module.train(self, mode)
        if mode:
            self.RCNN_base.stem.eval()

            def set_bn_eval(m):
                if m.__class__.__name__.find('BatchNorm') != -1:
                    m.eval()
            self.RCNN_base.apply(set_bn_eval)
            self.RCNN_top.apply(set_bn_eval)
        return self

    def _top_stride(self):
        return 2 if _opens_with_stride2_1x1(self.RCNN_top[0]) else 1

    def _head_to_tail(self, pool5, subsampled=False):
        if not subsampled:
            return self.RCNN_top(pool5).mean(3).mean(2)
        x = pool5
        if _TOP_NHWC and x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        for i, blk in enumerate(self.RCNN_top[0]):
            x = blk(x, subsampled=(i == 0))
        if _fmt(x) == torch.channels_last:
            # mean(3).mean(2) of the reference as ONE reduction over the 16 positions of a
            # channels-last row block (equal group sizes: the same mean, one rounding fewer)
            return x.permute(0, 2, 3, 1).reshape(x.size(0), -1, x.size(1)).mean(1)
        return x.mean(3).mean(2)


class resnet_coco(resnet):
    """The COCO-variant detector (lib/model/faster_rcnn/resnet_coatt_transformer_sk.py `resnet`,
    driven by trainval_net_coco.py:34 / test_net_coco.py:33).  Use with
    cfg_from_list(['ANCHOR_SCALES', [4, 8, 16, 32], 'MAX_NUM_GT_BOXES', 50])."""
    variant = 'coco'
