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
import ctypes
import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib, ops, system
from .config import cfg
from .roi_layers import ROIAlign
from .rpn import _ProposalTargetLayer, _RPN, _smooth_l1_loss
from .system import MultiHeadAttention, Transformer, _Linear, _split_k, conv2d_1x1


# ------------------------------------------------------------------------------------------
# channel block between AIT and layer4
# ------------------------------------------------------------------------------------------
# Test hooks, set from Python by the tests only (no environment switches: on a GPU the product path below is the one
# that runs, or it raises):
#   _SK_FULL     evaluate the SK block and the trunk's stage-closing blocks at EVERY position, as the reference does
#                (default: the dead positions are skipped -- same values, same gradients)
#   _TAIL_FUSED  the proposal tail (SK blocks + layer4 + mean) as ONE autograd node over ait_tail_fwd / ait_tail_bwd;
#                False = the nn.Module composition on PyTorch-ROCm convolutions, which the tests hold the node against
#   _TOP_NHWC / _BASE_NHWC / _ROI_NHWC   channels-last activations in the proposal tail / the C4 trunk / RoIAlign (the
#                AIT's token-major output IS channels-last); False = NCHW everywhere, the tests' reference configuration
_SK_FULL = False
_QUERY_SIDE_STREAM = True   # test hook: False = the query trunk on the step's own stream, behind the image's
_SIDE_STREAMS = {}


def _side_stream(device):
    key = torch.device(device).index
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
        # (the trunk's parameters receive gradients from both streams: intended, not a stale graph)
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
    return st

_TAIL_FUSED = True
_HEADS_KERNEL = True          # test hook: False = the two heads as torch nn.Linear calls (vendor GEMM)
_TRUNK_BF16 = True            # test hook: False = the C4 trunk in f32 also in the bf16 configuration (round 4's cfg5)
_AIT_OUT_BF16 = True          # test hook: False = the AIT's output stays f32 in the bf16 configuration (cast by the tail's entry)
_TAIL_BF16_MIOPEN = False     # test hook / A-B (scripts/ab_cfg5_tail.py): True = the bf16 configuration's proposal tail as the module
                              # composition on MIOpen's bf16 convolutions (round 5) instead of the library's bf16-storage node
_TOP_NHWC = True
_PAIR_GRADS = True            # test hook: False = a bottleneck's two input gradients summed by autograd (an add kernel each)
_BASE_NHWC = True
_ROI_NHWC = True


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
        if stride == 1:
            return conv(x)
        return F.conv2d(x, conv.weight, conv.bias, stride, conv.padding, conv.dilation, conv.groups)

    def forward(self, x, stride=1):
        """stride = 2 evaluates the block only at the even output positions (see
        _fasterRCNN.forward: the only consumer, layer4's stride-2 1x1 convolutions, never reads the
        others); the values at those positions are the same convolution sums."""
        """(the nn.Module composition: CPU tensors, and the reference point of the tests -- on the GPU the detector
        runs this block inside ait_tail_fwd / ait_tail_bwd, see _TailFn)"""
        if x.is_cuda and x.dtype == torch.float32 and self.n_state == 2 and x.numel() % 4 == 0:
            # convolutions on PyTorch-ROCm, then ReLU / square / branch sum in one fused HIP pass
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
                and not system._COATT_TORCH):
            return self._forward_hip(x_img, x_qry)
        if x_img.is_cuda and not system._COATT_TORCH:
            raise _lib.AitHipError("CoAttention: float32 features with 'division' normalisation expected on the GPU")
        ops.note_fallback("CoAttention", x_img)
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


def _alias(y):
    """a second tensor object over y's storage that autograd does not know as a view of y"""
    return torch.empty(0, dtype=y.dtype, device=y.device).set_(y.untyped_storage(), y.storage_offset(), y.size(), y.stride())


class _BnActPair(torch.autograd.Function):
    """_BnAct whose result is handed out TWICE -- two tensor objects over ONE storage: the first for the next block's
    convolution path, the second for its shortcut (resnet_sys_transformer_sk_dilat.py:89-107).  Autograd then hands the
    gradients of the two uses back SEPARATELY instead of adding them in front of this node (an add kernel per bottleneck:
    two reads and a write of the block's widest tensor, 91 launches a step), and the backward pass sums them while it streams
    them (ait_bn_act_bwd's second addend).  Same arithmetic: g = g_main + g_skip in f32, once."""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu, bf16):
        y = (ops.bn_act_fwd_bf16 if bf16 else ops.bn_act_fwd)(x, scale, shift, residual, relu)
        ctx.save_for_backward(y if relu else None, scale)
        ctx.relu, ctx.bf16 = relu, bf16
        ctx.has_res = residual is not None
        ctx.fmt = _fmt(x)
        ctx.set_materialize_grads(False)          # an unused output's gradient arrives as None, not as a tensor of zeros
        return y, _alias(y)

    @staticmethod
    def backward(ctx, g_main, g_skip):
        y, scale = ctx.saved_tensors
        gs = [g.contiguous(memory_format=ctx.fmt) for g in (g_main, g_skip) if g is not None]
        if not gs:
            return None, None, None, None, None, None
        bwd = ops.bn_act_bwd_bf16 if ctx.bf16 else ops.bn_act_bwd
        dx, dres = bwd(gs[0], y, scale, ctx.relu, ctx.has_res and ctx.needs_input_grad[3], dy2=gs[1] if len(gs) > 1 else None)
        return dx, None, None, dres, None, None


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


class _BnAct16(torch.autograd.Function):
    """_BnAct over bf16 channels-last tensors (the C4 trunk of the bf16 configuration)"""

    @staticmethod
    def forward(ctx, x, scale, shift, residual, relu):
        y = ops.bn_act_fwd_bf16(x, scale, shift, residual, relu)
        ctx.save_for_backward(y if relu else None, scale)
        ctx.relu = relu
        ctx.has_res = residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        y, scale = ctx.saved_tensors
        dy = dy.contiguous(memory_format=torch.channels_last)
        dx, dres = ops.bn_act_bwd_bf16(dy, y, scale, ctx.relu, ctx.has_res and ctx.needs_input_grad[3])
        return dx, None, None, dres, None


def bn_act(x, bn, residual=None, relu=True, pair=False):
    """relu(bn(x) + residual) for a FROZEN BatchNorm2d in eval mode (the only state the reference
    ever runs its BatchNorms in); anything else goes through torch.  pair=True: the result as TWO tensors over one storage
    (_BnActPair: for a consumer that reads it on two paths), or the same tensor twice where that node does not apply."""
    if _bn_frozen(bn) and x.is_cuda and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[1] % 8 == 0:
        scale, shift, _ = _bn_affine(bn)
        x = x.contiguous(memory_format=torch.channels_last)
        if residual is not None:
            residual = residual.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
        if pair:
            return _BnActPair.apply(x, scale, shift, residual, relu, True)
        return _BnAct16.apply(x, scale, shift, residual, relu)
    if not (_bn_frozen(bn) and x.is_cuda and x.dtype == torch.float32):
        y = bn(x)
        if residual is not None:
            y = y + residual
        y = F.relu(y) if relu else y
        return (y, y) if pair else y
    scale, shift, _ = _bn_affine(bn)
    fmt = _fmt(x)
    x = x.contiguous(memory_format=fmt)
    if residual is not None:
        residual = residual.contiguous(memory_format=fmt)
    if pair:
        return _BnActPair.apply(x, scale, shift, residual, relu, False)
    return _BnAct.apply(x, scale, shift, residual, relu)


def conv1x1_bn_act(x, conv, bn, residual=None, relu=True, stride1=False, pair=False):
    """relu(bn(conv(x)) + residual) for a bias-free 1x1 convolution.  `stride1`: run the convolution at stride 1
    whatever conv.stride says (the input is already subsampled).  The convolution is PyTorch-ROCm's (the C4 trunk,
    SURVEY 2), the frozen BN / residual / ReLU one HIP pass."""
    y = F.conv2d(x, conv.weight, None, (1, 1)) if stride1 else conv(x)
    return bn_act(y, bn, residual=residual, relu=relu, pair=pair)


# ------------------------------------------------------------------------------------------
# the proposal tail (both SK blocks + RCNN_top + mean over positions) as one autograd node
# ------------------------------------------------------------------------------------------
def _channels_last_weight(p):
    """conv weight [cout, cin/g, kh, kw] as the [cout][kh][kw][cin/g] matrix the library reads: the parameter is put
    into channels-last memory in place once (values, shape and state_dict unchanged)"""
    if not p.is_contiguous(memory_format=torch.channels_last):
        p.data = p.data.contiguous(memory_format=torch.channels_last)
    return p


def _only_plain_forward_hooks(mod):
    """True if every hook on `mod` is a plain forward hook (module, args, output) on the module itself -- the only kind the
    fused heads path can call by hand.  Forward pre-hooks, hooks registered with with_kwargs / always_call, backward
    (pre-)hooks, and hooks of any kind on a child module need torch's own dispatch."""
    def none(m, names):
        return not any(getattr(m, n, None) for n in names)
    if not none(mod, ("_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks", "_forward_hooks_with_kwargs",
                      "_forward_hooks_always_called")):
        return False
    for child in mod.modules():
        if child is not mod and not none(child, ("_forward_hooks", "_forward_pre_hooks", "_backward_hooks", "_backward_pre_hooks")):
            return False
    return True


class _HeadsFn(torch.autograd.Function):
    """(pooled proposal features [R, F], pooled query features [bs, F]) -> (bbox_pred [R, n_bbox], score [R, 2]) through
    ait_heads_fwd / ait_heads_bwd: RCNN_bbox_pred and the two Linears of RCNN_cls_score on the concatenation the
    reference builds (faster_rcnn_sys_transformer_sk_dilat.py:283-288), in the library (csrc/heads.hip)."""

    @staticmethod
    def forward(ctx, props, query, w_bbox, b_bbox, w1, b1, w2, b2):
        props, query = props.contiguous(), query.contiguous()
        bbox, hidden, score = ops.heads_fwd(props, query, w_bbox, b_bbox, w1, b1, w2, b2)
        ctx.save_for_backward(props, query, w_bbox, w1, w2, hidden)
        return bbox, score

    @staticmethod
    def backward(ctx, d_bbox, d_score):
        props, query, w_bbox, w1, w2, hidden = ctx.saved_tensors
        d_bbox = None if d_bbox is None else d_bbox.contiguous()
        d_score = None if d_score is None else d_score.contiguous()
        d_props, d_query, g = ops.heads_bwd(d_bbox, d_score, props, query, w_bbox, w1, w2, hidden,
                                            ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        return (d_props, d_query, g[0].view_as(w_bbox), g[1], g[2].view_as(w1), g[3], g[4].view_as(w2), g[5])


class _TailFn(torch.autograd.Function):
    """(AIT output tokens [bp*64, C], query tokens [bs*64, C]) -> pooled [bp + bs, 2048] through ait_tail_fwd; the
    backward (ait_tail_bwd) writes both input gradients and accumulates the 18 parameter gradients into one
    zero-filled buffer whose views are handed to autograd (include/ait_hip.h "The proposal tail")."""

    @staticmethod
    def forward(ctx, xp, xq, bp, bs, C, planes, n_blocks, W, keep, *params):
        L = _lib.lib()
        dev = xp.device
        xp, xq = xp.contiguous(), xq.contiguous()
        nbytes = int(L.ait_tail_saved_bytes(bp, bs, C, planes, n_blocks))
        if nbytes == 0:
            raise _lib.AitHipError("ait_tail: unsupported shape (bp=%d bs=%d C=%d planes=%d)" % (bp, bs, C, planes))
        saved = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        pooled = torch.empty((bp + bs, 4 * planes), dtype=torch.float32, device=dev)
        fmt = ctypes.c_uint(0)
        with torch.cuda.device(dev):
            ctx.flags = _lib.current_flags()      # (the backward runs in the product form of this forward: the layout of `saved`)
            rc = L.ait_tail_fwd(_lib.dev_ptr(xp), _lib.dev_ptr(xq), bp, bs, C, planes, n_blocks, ctypes.byref(W),
                                ctypes.c_void_p(saved.data_ptr()), nbytes, ctypes.byref(fmt), _lib.dev_ptr(pooled),
                                _lib.launch_ctx(dev, flags=ctx.flags), _lib.cur_stream(dev))
        _lib.check(rc, "ait_tail_fwd")
        ctx.fmt = fmt.value
        ctx.save_for_backward(xp, xq, saved)
        ctx.W, ctx.keep = W, keep
        ctx.cfg = (bp, bs, C, planes, n_blocks)
        ctx.meta = [tuple(t.shape) for t in params]
        ctx.strides = [tuple(t.stride()) for t in params]
        assert all(t.dim() != 4 or t.is_contiguous(memory_format=torch.channels_last) for t in params)
        return pooled

    @staticmethod
    def backward(ctx, d_pooled):
        L = _lib.lib()
        xp, xq, saved = ctx.saved_tensors
        bp, bs, C, planes, n_blocks = ctx.cfg
        dev = xp.device
        d_pooled = d_pooled.contiguous()
        sizes = [int(np.prod(sh)) for sh in ctx.meta]
        flat = torch.zeros(sum(sizes), dtype=torch.float32, device=dev)
        views, ptrs, o = [], [], 0
        for sh, st, n in zip(ctx.meta, ctx.strides, sizes):
            v = flat[o:o + n]
            ptrs.append(v.data_ptr())
            if len(sh) == 4:        # [cout][kh][kw][cin/g] memory = the channels-last strides of a [cout, cin/g, kh, kw] gradient
                # (spelled with the PARAMETER's own strides: for 1x1 kernels plain and channels-last strides describe the
                # same memory, and DDP's reducer aliases a gradient into its bucket only if the strides match literally)
                v = v.as_strided(sh, st)
            else:
                v = v.view(sh)
            views.append(v)
            o += n
        G = _lib.TailGrads()
        for j, name in enumerate(("sk_props", "sk_query")):
            g = getattr(G, name)
            g.w1, g.b1, g.w3, g.b3 = ptrs[4 * j:4 * j + 4]
        i = 8
        for k in range(n_blocks):
            g = G.block[k]
            g.conv1, g.conv2, g.conv3 = ptrs[i], ptrs[i + 1], ptrs[i + 2]
            i += 3
            if k == 0:
                g.down = ptrs[i]
                i += 1
        dxp = torch.empty_like(xp) if ctx.needs_input_grad[0] else None
        dxq = torch.empty_like(xq) if ctx.needs_input_grad[1] else None
        wbytes = int(L.ait_tail_bwd_workspace_bytes(bp, bs, C, planes, n_blocks))
        ws = torch.empty(wbytes, dtype=torch.uint8, device=dev)
        with torch.cuda.device(dev):
            rc = L.ait_tail_bwd(_lib.dev_ptr(d_pooled), _lib.dev_ptr(xp), _lib.dev_ptr(xq), bp, bs, C, planes, n_blocks,
                                ctypes.byref(ctx.W), ctypes.c_void_p(saved.data_ptr()), saved.numel(), ctx.fmt,
                                ctypes.c_void_p(ws.data_ptr()), wbytes,
                                None if dxp is None else _lib.dev_ptr(dxp), None if dxq is None else _lib.dev_ptr(dxq),
                                ctypes.byref(G), _lib.launch_ctx(dev, flags=ctx.flags), _lib.cur_stream(dev))
        _lib.check(rc, "ait_tail_bwd")
        return (dxp, dxq, None, None, None, None, None, None, None) + tuple(views)


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

    def forward(self, x, subsampled=False, out_stride=1, pair=False):
        """subsampled=True: `x` already holds only the positions this block's stride-s 1x1
        convolutions read (x[:, :, ::s, ::s]), so they run at stride 1.
        out_stride=s: the caller guarantees that this block's output is read ONLY by stride-s 1x1
        convolutions (the first block of the next stage); the block then produces just those
        positions -- conv2 runs at stride s (same 3x3 sums at the kept positions), conv3, the
        frozen BN, the residual and the ReLU are position-wise.
        x may be a PAIR (x_main, x_skip) of tensors over one storage (the previous block's pair=True result): the
        convolution path reads the first, the shortcut the second; pair=True returns this block's result as such a pair."""
        x, x_skip = x if isinstance(x, tuple) else (x, x)
        out = conv1x1_bn_act(x, self.conv1, self.bn1, stride1=subsampled)
        if out_stride == 1:
            out = bn_act(self.conv2(out), self.bn2)
        else:
            out = bn_act(F.conv2d(out, self.conv2.weight, None, out_stride, self.conv2.padding), self.bn2)
        if self.downsample is None:
            identity = x_skip if out_stride == 1 else _Subsample.apply(x_skip, out_stride)
        elif out_stride != 1:
            raise ValueError("out_stride needs an identity shortcut")
        else:
            identity = conv1x1_bn_act(x_skip, self.downsample[0], self.downsample[1], relu=False, stride1=subsampled)
        return conv1x1_bn_act(out, self.conv3, self.bn3, residual=identity, pair=pair)


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
    blocks = [(k, i, blk) for k, stage in enumerate(stages) for i, blk in enumerate(stage)]
    for n, (k, i, blk) in enumerate(blocks):
        stage = stages[k]
        nxt = stages[k + 1] if k + 1 < len(stages) else None
        skip = (not _SK_FULL) and nxt is not None and _opens_with_stride2_1x1(nxt) \
            and isinstance(stage[-1], Bottleneck) and stage[-1].downsample is None and len(stage) > 1
        # a block whose result the NEXT bottleneck reads on two paths (convolution + shortcut) hands it out as a pair
        # (_BnActPair): the two gradients come back separately and are summed inside the frozen-BN backward pass
        pair = _PAIR_GRADS and n + 1 < len(blocks) and isinstance(blk, Bottleneck) and isinstance(blocks[n + 1][2], Bottleneck)
        kw = {"pair": True} if pair else {}
        if isinstance(x, tuple) and not isinstance(blk, Bottleneck):
            x = x[0]
        x = blk(x, subsampled=(subsampled and i == 0), out_stride=2 if (skip and i == len(stage) - 1) else 1, **kw)
        if i == len(stage) - 1:
            subsampled = skip
    return x[0] if isinstance(x, tuple) else x


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
        if _TRUNK_BF16 and _BASE_NHWC and x.is_cuda and _lib.BF16_PRODUCTS:
            # the bf16 configuration (BASELINE configs[4], ops.set_matmul_dtype("bf16")): the trunk's convolutions run on
            # MIOpen with bf16 tensors (f32 accumulate; autocast casts the f32 master weights per call), the frozen-BN /
            # residual / ReLU passes between them read and write bf16 (ait_bn_act_*_bf16); the C4 feature goes back to f32
            # for the co-attention, the RPN and RoIAlign.  3-4x faster convolutions (profiles/r05_trunk_bf16.txt)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                x = self.stem[3](bn_act(self.stem[0](x), self.stem[1]))
                y = run_stages([self.layer1, self.layer2, self.layer3], x)
            return y.float(), None
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

        # (one program at every world size: the query trunk's launches are eager on the step's stream.  Round 3 replayed
        # them from HIP graphs on a side stream at N = 1 only -- 0 ms measured un-profiled, profiles/README_r03.md)
        if _QUERY_SIDE_STREAM and image.is_cuda:
            # the query patches' trip through the trunk is ~130 launches of a few microseconds each on 1/40 of the image's
            # pixels: on a second stream it runs in the shadow of the image's (autograd replays each node's backward on
            # the stream of its forward, so the backward overlaps the same way)
            cur = torch.cuda.current_stream(image.device)
            side = _side_stream(image.device)
            # the trunk's lazily filled device caches (the frozen BatchNorms' scale / shift) are filled HERE, on the
            # step's stream, in front of the fork: filled by the query pass on the side stream they would be read by
            # the image pass on `cur` with nothing ordering the reads behind the writes (first forward, and after any
            # change of a BatchNorm buffer such as load_state_dict)
            bns = self.__dict__.get("_trunk_bns")
            if bns is None:
                bns = self.__dict__["_trunk_bns"] = [m for m in self.RCNN_base.modules() if isinstance(m, nn.BatchNorm2d)]
            for m in bns:
                if _bn_frozen(m):
                    _bn_affine(m)
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                query_feat = self.RCNN_base(query)[0]             # [bs, 1024, 8, 8]
            image_feat, _ = self.RCNN_base(image)                 # [bs, 1024, H_i, W_i]
            cur.wait_stream(side)
            query_feat.record_stream(cur)
        else:
            image_feat, _ = self.RCNN_base(image)                 # [bs, 1024, H_i, W_i]
            query_feat = self.RCNN_base(query)[0]                 # [bs, 1024, 8, 8]
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
        # (A/B hook only: the MIOpen tail computes on bf16 tensors -- the AIT then hands its output over in bf16, sizes permitting)
        tail16 = bool(_TAIL_BF16_MIOPEN and _lib.BF16_PRODUCTS and props_feat.is_cuda and (1 if _SK_FULL else self._top_stride()) == 2)
        self.transformer.out_bf16 = tail16 and _AIT_OUT_BF16
        props_feat = self.transformer(x_props=props_feat, x_query=non_qry)   # [bs*P, 1024, 8, 8]
        # layer4 opens with stride-2 1x1 convolutions (Bottleneck.conv1 / downsample,
        # resnet_sys_transformer_sk_dilat.py:78,482-490): of the SK block's 8x8 output only the 16
        # even positions are ever read.  SK is position-wise after its convolutions, so it is
        # evaluated at those positions only (stride 2) and layer4 takes the result at stride 1:
        # same sums, same gradients (the dead positions receive exactly zero gradient in the
        # reference), 3/4 of the SK work not done.  AIT_SK_FULL=1 keeps the dead positions.
        sk_stride = 1 if _SK_FULL else self._top_stride()
        c_att = None
        if tail16:
            # (A/B hook _TAIL_BF16_MIOPEN: the bf16 configuration's proposal tail as the module composition on MIOpen with bf16
            # tensors, like the trunk -- round 5's path.  The product path is the branch below: ait_tail_* keeps layer4 on bf16
            # storage under AIT_CTX_BF16, csrc/tail.hip, at the same step time: profiles/r06_cfg5_tail_on_library.txt)
            with torch.autocast("cuda", dtype=torch.bfloat16):
                xp = props_feat.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                xq = non_qry.to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
                p16, q16 = self.sk(x_props=xp, x_query=xq, stride=sk_stride)
                props_feat = self._head_to_tail(p16, subsampled=True).float()
                query_feat = self._head_to_tail(q16, subsampled=True).float()
        elif self._tail_on_library(props_feat, non_qry, sk_stride):
            props_feat, query_feat = self._tail(props_feat, non_qry)                 # [bs*P, 2048], [bs, 2048]
        else:
            props_feat, query_feat = self.sk(x_props=props_feat, x_query=non_qry, stride=sk_stride)
            props_feat = self._head_to_tail(props_feat, subsampled=sk_stride != 1)   # [bs*P, 2048]
            query_feat = self._head_to_tail(query_feat, subsampled=sk_stride != 1)   # [bs, 2048]

        # (hooks the fused path cannot honour -- pre-hooks, kwargs / always-call hooks, backward hooks, any hook on a child
        # of RCNN_cls_score -- send the two heads through their modules, where torch dispatches them)
        hooks_plain = _only_plain_forward_hooks(self.RCNN_bbox_pred) and _only_plain_forward_hooks(self.RCNN_cls_score)
        if (props_feat.is_cuda and props_feat.dtype == torch.float32 and _HEADS_KERNEL and self.RCNN_bbox_pred.out_features <= 8
                and props_feat.shape[1] in (256, 512, 1024, 2048, 4096) and hooks_plain):
            # both heads in the library (csrc/heads.hip): no [bs*P, 4096] concatenation, no vendor GEMM under the logits
            bbox_pred, score = _HeadsFn.apply(props_feat, query_feat, self.RCNN_bbox_pred.weight, self.RCNN_bbox_pred.bias,
                                              self.RCNN_cls_score[0].weight, self.RCNN_cls_score[0].bias,
                                              self.RCNN_cls_score[1].weight, self.RCNN_cls_score[1].bias)
            # forward hooks registered on the two head modules still see their outputs (the reference's users read the
            # logits through a hook on RCNN_cls_score); the concatenated input does not exist here: args = ()
            for mod, name in ((self.RCNN_bbox_pred, "bbox"), (self.RCNN_cls_score, "score")):
                for hook in list(mod._forward_hooks.values()):
                    r = hook(mod, (), bbox_pred if name == "bbox" else score)
                    if r is not None:
                        if name == "bbox":
                            bbox_pred = r
                        else:
                            score = r
        else:
            if props_feat.is_cuda and _HEADS_KERNEL and hooks_plain:
                ops.note_fallback("heads", props_feat)
            bbox_pred = self.RCNN_bbox_pred(props_feat)
            stack_feat = torch.cat((props_feat.view(bs, num_props, -1),
                                    query_feat.unsqueeze(1).expand(-1, num_props, -1)), dim=2).reshape(-1, 4096)
            score = self.RCNN_cls_score(stack_feat)                          # similarity logits
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
        if _BASE_NHWC and self.variant == 'voc':
            # (the co-attention's query embedding is a MIOpen convolution on a channels-last feature: its weight gradient
            # comes back with channels-last strides, which DDP's reducer aliases only into a parameter spelled the same way)
            self.coattention.qry_emb.to(memory_format=torch.channels_last)


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

    def _tail_on_library(self, props, qry, sk_stride):
        """GPU tensors take the single-node tail (ait_tail_*); what it cannot express raises instead of silently
        running somewhere else.  CPU tensors (host-logic tests) take the module composition, counted as a fallback."""
        if not props.is_cuda:
            ops.note_fallback("proposal tail (SK + layer4)", props)
            return False
        if not _TAIL_FUSED or sk_stride != 2:
            return False                    # (test hooks: the module composition as the tests' reference point)
        blocks = list(self.RCNN_top[0])
        ok = (props.dtype == torch.float32 and qry.dtype == torch.float32 and props.size(2) == 8 and props.size(3) == 8
              and qry.shape[1:] == props.shape[1:] and props.size(1) % 1024 == 0 and 2 <= len(blocks) <= 4
              and all(isinstance(b, Bottleneck) and all(_bn_frozen(m) for m in (b.bn1, b.bn2, b.bn3)) for b in blocks)
              and blocks[0].downsample is not None and _bn_frozen(blocks[0].downsample[1])
              and all(b.downsample is None for b in blocks[1:]) and blocks[0].conv1.out_channels % 128 == 0
              and isinstance(self.sk.sk_props, SKBlock) and self.sk.sk_props.n_state == 2)
        if not ok:
            raise _lib.AitHipError("proposal tail: the library's ait_tail_* is built for 8x8 fp32 features, a frozen-BN "
                                   "Bottleneck layer4 and the two-branch SKBlock; got %s" % (tuple(props.shape),))
        return True

    def _tail_params(self):
        """the parameters in the order of ait_tail_grads (_TailFn.backward)"""
        ps = []
        for blk in (self.sk.sk_props, self.sk.sk_query):
            c1, c3 = blk.convs[0][0], blk.convs[1][0]
            ps += [c1.weight, c1.bias, _channels_last_weight(c3.weight), c3.bias]
        for k, b in enumerate(self.RCNN_top[0]):
            ps += [b.conv1.weight, _channels_last_weight(b.conv2.weight), b.conv3.weight]
            if k == 0:
                ps.append(b.downsample[0].weight)
        return ps

    def _tail_weights(self):
        """ait_tail_weights over this module's parameters and frozen-BN affines (pointers: rebuilt every call, cheap)"""
        keep = []

        def ptr(t):
            t = t.detach()
            if t.dim() == 4 and not (t.is_contiguous() or t.is_contiguous(memory_format=torch.channels_last)):
                raise _lib.AitHipError("convolution weight in an unexpected memory format")
            keep.append(t)
            return t.data_ptr()

        W = _lib.TailWeights()
        for name, blk in (("sk_props", self.sk.sk_props), ("sk_query", self.sk.sk_query)):
            c1, c3 = blk.convs[0][0], blk.convs[1][0]
            w = getattr(W, name)
            w.w1, w.b1, w.w3, w.b3 = ptr(c1.weight), ptr(c1.bias), ptr(_channels_last_weight(c3.weight)), ptr(c3.bias)
        for k, b in enumerate(self.RCNN_top[0]):
            w = W.block[k]
            w.conv1, w.conv2, w.conv3 = ptr(b.conv1.weight), ptr(_channels_last_weight(b.conv2.weight)), ptr(b.conv3.weight)
            for i, bn in ((1, b.bn1), (2, b.bn2), (3, b.bn3)):
                sc, sh, _ = _bn_affine(bn)
                setattr(w, "bn%d_scale" % i, ptr(sc))
                setattr(w, "bn%d_shift" % i, ptr(sh))
            if k == 0:
                w.down = ptr(b.downsample[0].weight)
                sc, sh, _ = _bn_affine(b.downsample[1])
                w.bnd_scale, w.bnd_shift = ptr(sc), ptr(sh)
        return W, keep

    def _tail(self, props_feat, non_qry):
        """props_feat [bp, C, 8, 8] (the AIT output: channels-last memory = token rows), non_qry [bs, C, 8, 8] ->
        (pooled proposals [bp, 2048], pooled queries [bs, 2048])"""
        bp, C = props_feat.size(0), props_feat.size(1)
        bs = non_qry.size(0)
        xp = props_feat.permute(0, 2, 3, 1).reshape(bp * 64, C)       # a view of channels-last memory
        xq = non_qry.permute(0, 2, 3, 1).reshape(bs * 64, C)
        blocks = self.RCNN_top[0]
        W, keep = self._tail_weights()
        pooled = _TailFn.apply(xp, xq, bp, bs, C, blocks[0].conv1.out_channels, len(blocks), W, keep, *self._tail_params())
        return pooled[:bp], pooled[bp:]

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
