"""Host-side mirror of `deeplio.models.nets` (nets/__init__.py:16-228 and the modules it wires):
same factory (`get_model(input_shape, cfg, device)`), same module/attribute names and therefore
the same state_dict keys and shapes, same forward contract
    model([[xyz, normals], imus]) -> (x_pos[B,S,3], x_ori[B,S,3]),
but every layer executes as hand-written gfx950 kernels through deeplio_amd.functional.

torch.nn.Conv2d / BatchNorm2d / Linear instances below are PARAMETER CONTAINERS only (they
give the reference's names, shapes and default initialisers); their forward is never called.
"""
import math
import os

import torch
import torch.nn as nn

from . import functional as Fh
from . import mixed
from . import ops
from .misc import get_config_container

# IMU branch backward scheduled at the fusion layer instead of at the end of the backward pass
_DEFER_IMU = True
# BatchNorm + ReLU of bypass-free Fire blocks applied by their consumer instead of being written (HISTORY 11)
_APPLY_ON_LOAD = os.environ.get("DLIO_APPLY_ON_LOAD", "1") != "0"
_STEM_AOL = True                                             # ... and of the stem by pool1


def _pair(v):
    return (v, v) if isinstance(v, int) else tuple(v)


class BaseNet(nn.Module):
    """base_net.py:6-30"""

    def __init__(self):
        super().__init__()
        self.pretrained = False
        self.output_shape = None

    def get_output_shape(self):
        return self.output_shape

    @property
    def name(self):
        return self.__class__.__name__.lower()

    @property
    def device(self):
        devices = ({p.device for p in self.parameters()} | {b.device for b in self.buffers()})
        if len(devices) != 1:
            raise RuntimeError('Cannot determine device: {} different devices found'.format(len(devices)))
        return next(iter(devices))

    def get_modules(self):
        return [self]


def _cbr(x, conv, bn, training, pre_relu=False, post_relu=True):
    return Fh.ConvBnAct.apply(x, conv.weight, conv.bias, bn.weight, bn.bias, bn.running_mean,
                              bn.running_var, _pair(conv.stride), _pair(conv.padding),
                              training, bn.momentum, bn.eps, pre_relu, post_relu)


def _pool_fast_behind(x, conv, pool):
    """does the pool behind `conv` applied to x take the 3x3 / pad 1 / stride (1|2, 2) kernels (ops.pool_fast_path)?"""
    kh, kw = _pair(conv.kernel_size); sh, sw = _pair(conv.stride); ph, pw = _pair(conv.padding)
    H = (x.shape[2] + 2 * ph - kh) // sh + 1
    W = (x.shape[3] + 2 * pw - kw) // sw + 1
    OH = ops.pool_out(H, pool.k, pool.stride[0], pool.pad[0], False)
    OW = ops.pool_out(W, pool.k, pool.stride[1], pool.pad[1], False)
    return ops.pool_fast_path(H, W, OH, OW, pool.k, pool.stride[0], pool.stride[1], pool.pad[0], pool.pad[1])


def _bump(bn, training):
    """nn.BatchNorm2d's num_batches_tracked bookkeeping.  Inside a DeepLIO model all counters
    are views of one int64 buffer bumped once per forward (share_bn_counters)."""
    if training and bn.num_batches_tracked is not None and not getattr(bn, "_shared_counter", False):
        bn.num_batches_tracked += 1


def share_bn_counters(net):
    bns = [m for m in net.modules() if isinstance(m, nn.BatchNorm2d) and m.num_batches_tracked is not None]
    if not bns:
        net._bn_counters = None
        return
    flat = torch.stack([m.num_batches_tracked.detach().reshape(()) for m in bns]).contiguous()
    for i, m in enumerate(bns):
        m._buffers['num_batches_tracked'] = flat[i]
        m._shared_counter = True
    net._bn_counters = flat


# ------------------------------------------------------------------------------ PointSeg
class Fire(nn.Module):
    """pointseg_modules.py:86-142"""

    def __init__(self, inplanes, squeeze_planes, expand1x1_planes, expand3x3_planes, bn=True,
                 bn_d=0.1, init='kaiming', bypass=None):
        super().__init__()
        if not bn:
            raise ValueError("Fire without batch norm is not on the DeepLIO path")
        self.squeeze = nn.Conv2d(inplanes, squeeze_planes, 1)
        self.squeeze_bn = nn.BatchNorm2d(squeeze_planes, momentum=bn_d)
        self.expand1x1 = nn.Conv2d(squeeze_planes, expand1x1_planes, 1)
        self.expand1x1_bn = nn.BatchNorm2d(expand1x1_planes, momentum=bn_d)
        self.expand3x3 = nn.Conv2d(squeeze_planes, expand3x3_planes, 3, padding=1)
        self.expand3x3_bn = nn.BatchNorm2d(expand3x3_planes, momentum=bn_d)
        same = inplanes == expand1x1_planes + expand3x3_planes
        # pointseg_modules.py:108-112: 'complex' gives blocks whose width changes a 1x1 "upsample"
        # convolution of the input as residual (and blocks whose width does not change NO residual,
        # :135-140); 'simple' adds the input where the width allows
        if bypass == "complex" and not same:
            self.upsample = nn.Conv2d(inplanes, expand1x1_planes + expand3x3_planes, 1)
        self.residual = bypass == "simple" and same

    def forward(self, x, want_gap=False, defer=False, pool=None):
        """want_gap: also return the [N, C] plane averages of the output (for a following SELayer).
        x may be a (raw, aff) pair from a deferring block (functional.FireFn); defer: return such a pair.
        pool (k, stride, pad): SELayer + MaxPool2d(pool) are the ONLY readers of this block's output -- the block may then
        return (pooled maximum, plane averages, arg-max map) instead (functional.FireFn, poolfuse)."""
        tr = self.training
        s, sb, e1, e1b, e3, e3b = (self.squeeze, self.squeeze_bn, self.expand1x1, self.expand1x1_bn,
                                   self.expand3x3, self.expand3x3_bn)
        for b in (sb, e1b, e3b):
            _bump(b, tr)
        x_aff = None
        if isinstance(x, tuple):
            x, x_aff = x
        up = getattr(self, "upsample", None)
        bf16 = x.dtype == torch.bfloat16          # mixed-precision region (deeplio_amd.mixed): dispatch on the input
        if bf16 and up is not None:
            raise ValueError("Fire bypass 'complex' has no bf16 kernel (mixed precision supports None / 'simple')")
        args = (x, s.weight, s.bias, sb.weight, sb.bias, sb.running_mean, sb.running_var,
                e1.weight, e1.bias, e1b.weight, e1b.bias, e1b.running_mean, e1b.running_var,
                e3.weight, e3.bias, e3b.weight, e3b.bias, e3b.running_mean, e3b.running_var,
                tr, sb.momentum, sb.eps, self.residual, want_gap and up is None)
        if bf16:
            out = mixed.FireFn.apply(*args)
        else:
            out = Fh.FireFn.apply(*args, x_aff, defer, pool if up is None else None)
        if up is None:
            return out
        out = Fh.ConvAddFn.apply(x, up.weight, up.bias, out)      # out + upsample(identity), :136-138
        return (out, None) if want_gap else out                    # the SELayer takes its own averages

    def can_defer(self):
        """this block's activated output may stay unwritten (apply-on-load in the next Fire block): training,
        no bypass / upsample on THIS block's output path -- and the plane-structured BatchNorm kernels in use: only
        they take the residual's (mean, scale, beta) (with DLIO_PLANE_BN=0 the consumer would add the RAW tensor)"""
        return (self.training and Fh._PLANE_BN[0] and not self.residual and getattr(self, "upsample", None) is None)


class SELayer(nn.Module):
    """pointseg_modules.py:203-221; `pool` fuses the following MaxPool2d."""

    def __init__(self, in_features, reduction=16):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(in_features, in_features // reduction, bias=False), nn.ReLU(),
                                nn.Linear(in_features // reduction, in_features, bias=False), nn.Sigmoid())

    def forward(self, x, pool=None, gap=None, pooled=None):
        if x.dtype == torch.bfloat16:
            if pool is None:
                raise ValueError("the bf16 SELayer exists fused with the max-pool behind it only (as PSEncoder uses it)")
            return mixed.SEPoolFn.apply(x, self.fc[0].weight, self.fc[2].weight, pool, gap)
        return Fh.SEPoolFn.apply(x, self.fc[0].weight, self.fc[2].weight, pool, gap, pooled)


class MaxPool(nn.Module):
    def __init__(self, k, stride, pad, ceil_mode=False):
        super().__init__()
        self.k, self.stride, self.pad, self.ceil_mode = k, _pair(stride), _pair(pad), ceil_mode

    def forward(self, x):
        return Fh.MaxPoolFn.apply(x, self.k, self.stride, self.pad, self.ceil_mode)


PS_BLOCKS = (  # pointseg_net.py:24-55
    ("fire_blk1", ((64, 16, 64), (128, 16, 64)), 128, (1, 2)),
    ("fire_blk2", ((128, 32, 128), (256, 32, 128)), 256, (1, 2)),
    ("fire_blk3", ((256, 48, 192), (384, 48, 192), (384, 64, 256), (512, 64, 256)), 512, (2, 2)),
    ("fire_blk4", ((512, 64, 256), (512, 64, 256)), 512, (2, 2)),
    ("fire_blk5", ((512, 80, 384), (768, 80, 384)), None, None),
)


class PSEncoder(BaseNet):
    """pointseg_net.py:9-82; output shape by arithmetic instead of a dummy forward."""

    def __init__(self, input_shape, cfg, bn_d=0.1):
        super().__init__()
        self.bypass = cfg['bypass']
        # 'bf16': activations / activation gradients of the Fire blocks are stored in bf16 (BASELINE
        # configs[4], deeplio_amd.mixed); parameters, statistics and the stem stay fp32
        self.precision = str(cfg.get('precision', 'fp32')).lower()
        if self.precision not in ('fp32', 'bf16'):
            raise ValueError("Wrong precision {} (fp32 or bf16)".format(self.precision))
        self.input_shape = input_shape
        c, h, w = input_shape
        self.conv1a = nn.Sequential(nn.Conv2d(c, 64, (3, 5), (1, 2), (1, 2)),
                                    nn.BatchNorm2d(64, momentum=bn_d), nn.ReLU(inplace=True))
        self.pool1 = MaxPool(3, (1, 2), 1)
        hh, ww = h, (w + 4 - 5) // 2 + 1
        ww = (ww + 2 - 3) // 2 + 1
        for name, fires, se, pool in PS_BLOCKS:
            mods = []
            for i, (ci, sq, e) in enumerate(fires):
                last = name == "fire_blk5" and i == len(fires) - 1
                mods.append(Fire(ci, sq, e, e, bn=True, bn_d=bn_d, bypass=None if last else self.bypass))
            if se:
                mods.append(SELayer(se, reduction=2))
            if pool:
                mods.append(MaxPool(3, pool, 1))
                hh, ww = (hh + 2 - 3) // pool[0] + 1, (ww + 2 - 3) // pool[1] + 1
            setattr(self, name, nn.Sequential(*mods))
        self.output_shapes = torch.Size([1, 768, hh, ww])

    def forward(self, x):
        for x in self.forward_steps(x):
            pass
        return x

    def forward_steps(self, x):
        """the forward pass as a generator that yields after every module: the two siamese encoders
        are issued in lockstep on their two streams (BaseLidarFeatNet.forward), so neither stream
        waits while the host issues the other encoder -- and, because autograd replays nodes in
        reverse creation order, backward alternates between the two streams as well"""
        tr = self.training
        _bump(self.conv1a[1], tr)
        p1, c1, b1 = self.pool1, self.conv1a[0], self.conv1a[1]
        if tr and _APPLY_ON_LOAD and _STEM_AOL and x.is_cuda and not p1.ceil_mode and _pool_fast_behind(x, c1, p1):
            # the stem's BatchNorm + ReLU is applied by pool1 while it loads (apply-on-load)
            x = Fh.ConvBnActPoolFn.apply(x, c1.weight, c1.bias, b1.weight, b1.bias, b1.running_mean, b1.running_var,
                                         _pair(c1.stride), _pair(c1.padding), b1.momentum, b1.eps, p1.k, p1.stride, p1.pad)
        else:
            x = p1(_cbr(x, c1, b1, tr))
        if self.precision == 'bf16':
            x = mixed.CastFn.apply(x)
        yield x
        for name, *_ in PS_BLOCKS:
            mods = list(getattr(self, name))
            i = 0
            gap = pooled = None
            while i < len(mods):
                m = mods[i]
                if isinstance(m, SELayer) and i + 1 < len(mods) and isinstance(mods[i + 1], MaxPool):
                    p = mods[i + 1]
                    x = m(x, (p.k, p.stride, p.pad), gap, pooled)
                    i += 2
                elif isinstance(m, SELayer):
                    x = m(x, None, gap)
                    i += 1
                elif isinstance(m, Fire) and i + 1 < len(mods) and isinstance(mods[i + 1], SELayer):
                    # the SELayer's squeeze comes out of the BN apply; with the pool behind it the block may pool its own
                    # output (its only readers are these two modules)
                    p = mods[i + 2] if (i + 2 < len(mods) and isinstance(mods[i + 2], MaxPool) and not mods[i + 2].ceil_mode) else None
                    r = m(x, want_gap=True, pool=(p.k, p.stride, p.pad) if (p is not None and self.precision == 'fp32') else None)
                    x, gap = r[0], r[1]
                    pooled = getattr(x, "_dlio_pooled", None) if len(r) == 3 else None
                    i += 1
                elif (isinstance(m, Fire) and i + 1 < len(mods) and isinstance(mods[i + 1], Fire) and _APPLY_ON_LOAD
                      and m.can_defer() and getattr(mods[i + 1], "upsample", None) is None
                      and self.precision == 'fp32'):
                    # the only consumer is the next Fire block: it applies this block's BatchNorm + ReLU
                    # while it loads (squeeze convolution, bypass residual, squeeze weight gradient)
                    x = m(x, defer=True)
                    i += 1
                else:
                    x = m(x)
                    i += 1
                yield x

    def get_output_shape(self):
        return self.output_shapes


# ------------------------------------------------------------------------------ FlowNet / ResNet / Simple-1
def conv(batch_norm, in_planes, out_planes, kernel_size=(3, 3), stride=1):
    """base_net.py:55-71 (container only)"""
    if not batch_norm:
        raise ValueError("conv() without batch norm is not on the DeepLIO path")
    pad = ((kernel_size[0] - 1) // 2, (kernel_size[1] - 1) // 2)
    return nn.Sequential(nn.Conv2d(in_planes, out_planes, kernel_size, stride, pad, bias=False),
                         nn.BatchNorm2d(out_planes), nn.ReLU())


FLOWNET_LAYERS = (("conv1", None, 64, (5, 7), (1, 2)), ("conv2", 64, 128, (3, 5), (1, 2)),
                  ("conv3", 128, 256, (3, 5), (1, 2)), ("conv3_1", 256, 256, (3, 3), 1),
                  ("conv4", 256, 512, (3, 3), 2), ("conv4_1", 512, 512, (3, 3), 1),
                  ("conv5", 512, 512, (3, 3), 2), ("conv5_1", 512, 512, (3, 3), 1),
                  ("conv6", 512, 1024, (3, 3), 2))


class FlowNetEncoder(nn.Module):
    """lidar_feat_nets.py:240-267"""
    out_channels = 1024

    def __init__(self, input_shape, batch_norm=True):
        super().__init__()
        c = input_shape[0]
        for name, ci, co, k, s in FLOWNET_LAYERS:
            setattr(self, name, conv(batch_norm, c if ci is None else ci, co, k, s))

    def forward(self, x):
        tr = self.training
        for name, *_ in FLOWNET_LAYERS:
            seq = getattr(self, name)
            _bump(seq[1], tr)
            x = _cbr(x, seq[0], seq[1], tr)
        return Fh.GapFn.apply(x)


class BasicBlock(nn.Module):
    """torchvision BasicBlock semantics (used by resnet.py:3,86-92)"""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        tr = self.training
        for b in (self.bn1, self.bn2):
            _bump(b, tr)
        idt = x
        if self.downsample is not None:
            _bump(self.downsample[1], tr)
            idt = _cbr(x, self.downsample[0], self.downsample[1], tr, post_relu=False)
        out = _cbr(x, self.conv1, self.bn1, tr)
        out = _cbr(out, self.conv2, self.bn2, tr, post_relu=False)
        return Fh.BinaryFn.apply(out, idt, 3)


class ResNetEncoder(nn.Module):
    """resnet.py:14-112, layers [3,3,3,2]"""
    out_channels = 512
    STAGES = ((64, 3, (1, 2)), (128, 3, (1, 2)), (256, 3, (2, 2)), (512, 2, (2, 2)))

    def __init__(self, input_shape):
        super().__init__()
        c = input_shape[0]
        self.conv1 = nn.Conv2d(c, 64, (5, 7), (1, 1), (2, 3), bias=True)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = MaxPool(3, (1, 2), (1, 1))
        inpl = 64
        for i, (planes, nblk, stride) in enumerate(self.STAGES):
            down = nn.Sequential(nn.Conv2d(inpl, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
            blocks = [BasicBlock(inpl, planes, stride, down)]
            blocks += [BasicBlock(planes, planes) for _ in range(nblk - 1)]
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
            inpl = planes
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)

    def forward(self, x):
        tr = self.training
        _bump(self.bn1, tr)
        x = self.maxpool(_cbr(x, self.conv1, self.bn1, tr))
        for i in range(4):
            for blk in getattr(self, "layer%d" % (i + 1)):
                x = blk(x)
        return Fh.GapFn.apply(x)


SIMPLE_LAYERS = ((1, None, 64, (5, 7), (1, 2), (2, 3), (1, 2)), (2, 64, 128, (3, 5), (1, 1), (1, 2), (1, 2)),
                 (3, 128, 128, 3, (1, 1), 1, None), (4, 128, 256, 3, (1, 1), 1, (2, 2)),
                 (5, 256, 256, 3, (1, 1), 1, None), (6, 256, 512, 3, (1, 1), 1, (2, 2)),
                 (7, 512, 512, 3, (1, 1), 1, None))


class FeatureNetSimple1(nn.Module):
    """lidar_feat_nets.py:270-342 (conv -> ReLU -> BN, ceil-mode pools)"""
    out_channels = 512

    def __init__(self, input_shape, bypass=False):
        super().__init__()
        if bypass:
            raise ValueError("lidar-feat-simple-1 with bypass=true has no reference behaviour "
                             "(the reference crashes in forward)")
        c = input_shape[0]
        for i, ci, co, k, s, p, pool in SIMPLE_LAYERS:
            setattr(self, "conv%d" % i, nn.Conv2d(c if ci is None else ci, co, k, s, p))
            setattr(self, "bn%d" % i, nn.BatchNorm2d(co))
            if pool:
                setattr(self, "pool%d" % i, MaxPool(3, pool, (1, 1), ceil_mode=True))

    def forward(self, x):
        tr = self.training
        for i, _, _, _, _, _, pool in SIMPLE_LAYERS:
            bn = getattr(self, "bn%d" % i)
            _bump(bn, tr)
            x = _cbr(x, getattr(self, "conv%d" % i), bn, tr, pre_relu=True, post_relu=False)
            if pool:
                x = getattr(self, "pool%d" % i)(x)
        return Fh.GapFn.apply(x)


# ------------------------------------------------------------------------------ lidar feature nets
def _gap(x):
    """global average pool -> fp32 [N, C] (leaves the bf16 region of a mixed-precision encoder)"""
    return (mixed.GapFn if x.dtype == torch.bfloat16 else Fh.GapFn).apply(x)


class BaseLidarFeatNet(BaseNet):
    """lidar_feat_nets.py:12-43 + the shared two-stream forward of :73-237."""
    act = ops.ACT_RELU
    drop_before_fc = False

    def __init__(self, input_shape, cfg):
        super().__init__()
        self.p = cfg['dropout']
        self.fusion = cfg['fusion']
        cc = get_config_container()
        self.seq_size, self.timestamps, self.combinations = cc.seq_size, cc.timestamps, cc.combinations
        self.input_shape = input_shape
        c, h, w = input_shape
        self.encoder1 = self.make_encoder((2 * c, h, w), cfg)
        self.encoder2 = self.make_encoder((2 * c, h, w), cfg)
        nfeat = self.encoder1.out_channels * (2 if self.fusion == 'cat' else 1)
        self.fc1 = nn.Linear(nfeat, 128)
        self.output_shape = torch.Size([1, self.seq_size, 128])
        self.two_streams = os.environ.get("DLIO_TWO_STREAMS", "1") != "0"
        self.interleave = True
        self._side = None

    def forward(self, x):
        return self.head(*self.encode(x))

    def encode(self, x):
        """the two siamese encoders -> (feature a, feature b, (batch, sequence))"""
        xyz, nrm = x[0], x[1]
        b, s, t, c, h, w = xyz.shape
        xa, xb = xyz.reshape(b * s, t * c, h, w), nrm.reshape(b * s, t * c, h, w)
        if xyz.is_cuda and self.two_streams:
            # The two encoders are independent until the feature fusion: run the normals stream
            # on a second HIP stream so that its HBM-bound kernels (BN, pools) overlap the
            # MFMA-bound convolutions of the xyz stream and small layers fill idle CUs.
            if self._side is None:
                self._side = Fh.aux_stream(xyz.device, "encoder2")
            main = torch.cuda.current_stream()
            self._side.wait_stream(main)
            if self.interleave and hasattr(self.encoder1, "forward_steps"):
                ga, gb = self.encoder1.forward_steps(xa), self.encoder2.forward_steps(xb)
                fa = fb = None
                while ga is not None or gb is not None:
                    if gb is not None:
                        with Fh.on_stream(self._side):
                            try:
                                fb = next(gb)
                            except StopIteration:
                                gb = None
                    if ga is not None:
                        try:
                            fa = next(ga)
                        except StopIteration:
                            ga = None
                fused = self._fused_head(fa, fb)
                if fb.dim() == 4 and not fused:
                    with Fh.on_stream(self._side):
                        fb = _gap(fb)
            else:
                with Fh.on_stream(self._side):
                    fb = self.encoder2(xb)
                fa = self.encoder1(xa)
                fused = self._fused_head(fa, fb)
                if fb.dim() == 4 and not fused:
                    with Fh.on_stream(self._side):
                        fb = _gap(fb)
            if fa.dim() == 4 and not fused:
                fa = _gap(fa)
            main.wait_stream(self._side)
            fb.record_stream(main)
        else:
            fa, fb = self.encoder1(xa), self.encoder2(xb)
            if fa.dim() == 4 and not self._fused_head(fa, fb):
                fa, fb = _gap(fa), _gap(fb)
        return fa, fb, (b, s)

    def _fused_head(self, fa, fb):
        """the encoder outputs go to head() as maps: plane averages, add / sub, fc1 and its activation are one launch
        (functional.PairFuseFcFn) -- fp32 maps, fusion 'add' / 'sub', the dropout behind fc1"""
        return (Fh._PAIR_FUSE[0] and fa.dim() == 4 and fa.is_cuda and fa.dtype == torch.float32 and fb.dtype == torch.float32
                and self.fusion in ('add', 'sub') and not self.drop_before_fc and fa.shape == fb.shape)

    def head(self, fa, fb, bs):
        """fusion of the two encoder features, fc1, dropout (the part behind the encoders: small, serial)"""
        b, s = bs
        if fa.dim() == 4:       # maps (see _fused_head)
            y = Fh.PairFuseFcFn.apply(fa, fb, 0 if self.fusion == 'add' else 1, self.fc1.weight, self.fc1.bias, self.act)
            return Fh.dropout(y, self.p, self.training).view(b, s, -1)
        if self.fusion == 'cat':
            y = Fh.Cat2Fn.apply(fa, fb)
        else:
            y = Fh.BinaryFn.apply(fa, fb, 0 if self.fusion == 'add' else 1)
        if self.drop_before_fc:
            y = Fh.dropout(y, self.p, self.training)
            y = Fh.LinearFn.apply(y, self.fc1.weight, self.fc1.bias, self.act)
        else:
            y = Fh.LinearFn.apply(y, self.fc1.weight, self.fc1.bias, self.act)
            y = Fh.dropout(y, self.p, self.training)
        return y.view(b, s, -1)


class LidarPointSegFeat(BaseLidarFeatNet):
    """lidar_feat_nets.py:46-101"""

    def make_encoder(self, shape, cfg):
        self.part = cfg['part'].lower()
        enc = PSEncoder(shape, cfg)
        enc.out_channels = 768
        return enc


class LidarFlowNetFeat(BaseLidarFeatNet):
    """lidar_feat_nets.py:104-148"""

    def make_encoder(self, shape, cfg):
        return FlowNetEncoder(list(shape))


class LidarResNetFeat(BaseLidarFeatNet):
    """lidar_feat_nets.py:151-189 (dropout before fc1)"""
    drop_before_fc = True

    def make_encoder(self, shape, cfg):
        return ResNetEncoder(list(shape))


class LidarSimpleFeat1(BaseLidarFeatNet):
    """lidar_feat_nets.py:192-237 (dropout, fc1, leaky-ReLU)"""
    drop_before_fc = True
    act = ops.ACT_LEAKY

    def make_encoder(self, shape, cfg):
        return FeatureNetSimple1(list(shape), bypass=cfg['bypass'])


# ------------------------------------------------------------------------------ IMU nets
class BaseImuFeatNet(BaseNet):
    def __init__(self, cfg):
        super().__init__()
        self.p = cfg['dropout']
        self.input_size = cfg['input-size']
        self.num_layers = cfg.get('num-layers', 2)
        cc = get_config_container()
        self.seq_size, self.combinations = cc.seq_size, cc.combinations


class ImuFeatFC(BaseImuFeatNet):
    """imu_feat_nets.py:21-53: all B*S*T samples go through the MLP as one skinny GEMM chain,
    then a segmented sum over T (the reference loops over b, s in Python)."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.hidden_size = cfg.get('hidden-size', [6, 6])
        self.num_layers = len(self.hidden_size)
        dims = [self.input_size] + list(self.hidden_size)
        self.net = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        self.output_shape = [1, self.seq_size, self.hidden_size[-1]]

    def forward(self, x):
        if not torch.is_tensor(x):
            x = torch.stack([torch.stack(list(xb)) for xb in x])
        b, s, t, n = x.shape
        y = x.reshape(b * s * t, n)
        for m in self.net:
            y = Fh.LinearFn.apply(y, m.weight, m.bias, ops.ACT_LEAKY)
        y = Fh.dropout(y, self.p, self.training)
        return Fh.SegSumFn.apply(y.view(b * s, t, -1)).view(b, self.seq_size, -1)


class RNNParams(nn.Module):
    """Parameter container with nn.LSTM / nn.GRU names, shapes and default init
    (U(-1/sqrt(H), 1/sqrt(H))): weight_ih_l{k}[_reverse], weight_hh_l{k}[_reverse], bias_*."""

    def __init__(self, mode, input_size, hidden_size, num_layers, bidirectional, dropout):
        super().__init__()
        self.mode, self.input_size, self.hidden_size = mode, input_size, hidden_size
        self.num_layers, self.bidirectional, self.dropout = num_layers, bidirectional, float(dropout)
        G = 4 if mode == "lstm" else 3
        D = 2 if bidirectional else 1
        k = 1.0 / math.sqrt(hidden_size)
        self._order = []
        for l in range(num_layers):
            for d in range(D):
                sfx = "_l%d%s" % (l, "_reverse" if d == 1 else "")
                kin = input_size if l == 0 else D * hidden_size
                for nm, shape in (("weight_ih", (G * hidden_size, kin)), ("weight_hh", (G * hidden_size, hidden_size)),
                                  ("bias_ih", (G * hidden_size,)), ("bias_hh", (G * hidden_size,))):
                    p = nn.Parameter(torch.empty(shape).uniform_(-k, k))
                    self.register_parameter(nm + sfx, p)
                    self._order.append(nm + sfx)

    def flat_weights(self):
        return [getattr(self, n) for n in self._order]

    def run(self, x4, training, top_fwd_only=False):
        """x4 [B, Sg, T, I] -> [B, Sg, T, D*H] (top_fwd_only: the caller keeps [..., :H] only -- the streamed layer path then
        returns [B, Sg, T, H] and never runs the top layer's reverse direction; the other paths ignore the hint)"""
        D = 2 if self.bidirectional else 1
        if Fh.lstm_stack_ok(x4, self.mode, self.hidden_size, self.num_layers, D):
            # one sequence per sample, wide hidden state (the odometry net): one both-directions launch sequence per layer
            return Fh.LstmStackFn.apply(x4[:, 0], self.hidden_size, self.num_layers, D, self.dropout, training,
                                        bool(top_fwd_only and D == 2), *self.flat_weights()).unsqueeze(1)
        return Fh.RNNFn.apply(x4, self.mode, self.hidden_size, self.num_layers, D, self.dropout,
                              training, *self.flat_weights())


def _make_rnn(cfg, input_size, num_layers, p):
    mode = 'gru' if cfg['type'].lower() == 'gru' else 'lstm'
    return RNNParams(mode, input_size, cfg.get('hidden-size', 6), num_layers,
                     cfg.get('bidirectional', False), p)


class ImufeatRNN0(BaseImuFeatNet):
    """imu_feat_nets.py:56-83: state carried across the S sub-sequences; feature = last step of
    the forward direction of the top layer."""

    def __init__(self, cfg):
        super().__init__(cfg)
        self.hidden_size = cfg.get('hidden-size', 6)
        self.bidirectional = cfg.get('bidirectional', False)
        self.rnn = _make_rnn(cfg, self.input_size, self.num_layers, self.p)
        self.num_dir = 2 if self.bidirectional else 1
        self.output_shape = [1, self.seq_size, self.hidden_size]

    def forward(self, x):
        tops = self.rnn.run(x, self.training)          # [B, S, T, D*H]
        return tops[:, :, -1, :self.hidden_size]


# ------------------------------------------------------------------------------ fusion / odometry
class DeepLIOFusionCat:
    """fusion_nets.py:9-37 (plain class: no parameters, not in the state_dict)"""
    name = "deepliofusioncat"

    def __init__(self, input_shapes, cfg):
        cc = get_config_container()
        self.seq_size, self.combinations = cc.seq_size, cc.combinations
        self.type = cfg.get('type', 'cat').lower()
        if self.type != 'cat':
            raise NotImplementedError()
        self.input_shapes = input_shapes
        self.output_shape = [1, self.seq_size, sum(s[-1] for s in input_shapes)]

    def forward(self, x):
        return Fh.Cat2Fn.apply(x[0], x[1])

    def get_output_shape(self):
        return self.output_shape

    def __call__(self, x):
        return self.forward(x)


class DeepLIOFusionSoft(BaseNet):
    """fusion_nets.py:40-78.  The gates multiply out of place (same values; the reference's
    in-place `*=` on its inputs breaks autograd for ReLU-terminated features)."""

    def __init__(self, input_shapes, cfg):
        super().__init__()
        cc = get_config_container()
        self.seq_size, self.combinations = cc.seq_size, cc.combinations
        self.input_shapes = input_shapes
        self.s1_feat = self.s2_feat = None
        total = sum(s[-1] for s in input_shapes)
        self.layers = nn.ModuleList([nn.Linear(total, s[-1]) for s in input_shapes])
        self.output_shape = [1, self.seq_size, total]

    def forward(self, x):
        lidar, imu = x[0], x[1]
        if (Fh._TAIL_FUSED[0] and lidar.is_cuda and lidar.dtype == torch.float32 and imu.dtype == torch.float32
                and lidar.shape[:-1] == imu.shape[:-1]
                and ops.soft_fusion_ok(lidar.numel() // lidar.shape[-1], lidar.shape[-1], imu.shape[-1])):
            # cat + two gated linear layers + two products + cat as one launch
            out, gate = Fh.SoftFusionFn.apply(lidar, imu, self.layers[0].weight, self.layers[0].bias, self.layers[1].weight,
                                              self.layers[1].bias)
            fa = lidar.shape[-1]
            self.s1_feat, self.s2_feat = gate[..., :fa], gate[..., fa:]
            return out
        cat = Fh.Cat2Fn.apply(lidar, imu)
        self.s1_feat = Fh.LinearFn.apply(cat, self.layers[0].weight, self.layers[0].bias, ops.ACT_SIGMOID)
        self.s2_feat = Fh.LinearFn.apply(cat, self.layers[1].weight, self.layers[1].bias, ops.ACT_SIGMOID)
        return Fh.Cat2Fn.apply(Fh.BinaryFn.apply(lidar, self.s1_feat, 2),
                               Fh.BinaryFn.apply(imu, self.s2_feat, 2))


class OdomFeatFC(BaseNet):
    """odom_feat_nets.py:8-45 (key 'hidden-size'; config.yaml's 'size' is ignored)"""

    def __init__(self, in_features, cfg):
        super().__init__()
        self.input_size = in_features
        self.hidden_size = cfg.get('hidden-size', [256, 128])
        self.p = cfg.get('dropout', 0.)
        cc = get_config_container()
        self.seq_size, self.combinations = cc.seq_size, cc.combinations
        dims = [in_features] + list(self.hidden_size)
        self.layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])

    def forward(self, x):
        b, s, n = x.shape
        y = x.reshape(b * s, n)
        for layer in self.layers:
            y = Fh.LinearFn.apply(y, layer.weight, layer.bias, ops.ACT_LEAKY)
        y = Fh.dropout(y, self.p, self.training)
        return y.view(b, s, -1)

    def get_output_shape(self):
        return [1, 1, self.hidden_size[-1]]


class OdomFeatRNN(BaseNet):
    """odom_feat_nets.py:48-86: RNN over the S axis, forward-direction half kept."""

    def __init__(self, in_features, cfg):
        super().__init__()
        self.hidden_size = cfg.get('hidden-size', 6)
        self.p = cfg.get('dropout', 0.)
        self.bidirectional = cfg.get('bidirectional', False)
        self.input_size = in_features
        cc = get_config_container()
        self.seq_size, self.combinations = cc.seq_size, cc.combinations
        self.rnn = _make_rnn(cfg, in_features, cfg.get('num-layers', 2), self.p)
        self.num_dir = 2 if self.bidirectional else 1

    def forward(self, x):
        return self.forward_full(x)[:, :, :self.hidden_size]

    def forward_full(self, x):
        """[B, S, H | D H]: the top layer's outputs whose first H columns the model keeps (:82) -- the streamed layer path
        delivers just those, the per-direction path both directions"""
        b, s, n = x.shape
        r = self.rnn
        D = 2 if r.bidirectional else 1
        if Fh.lstm_stack_ok(x.unsqueeze(1), r.mode, r.hidden_size, r.num_layers, D):
            # (no unsqueeze / select views around the node: their backward is a zero fill + a copy each)
            return Fh.LstmStackFn.apply(x, r.hidden_size, r.num_layers, D, r.dropout, self.training, D == 2, *r.flat_weights())
        tops = r.run(x.reshape(b, 1, s, n), self.training, top_fwd_only=True)   # [B, 1, S, H | D*H]
        return tops[:, 0]

    def get_output_shape(self):
        return [1, 1, self.hidden_size]


# ------------------------------------------------------------------------------ top module
class DeepLIO(BaseNet):
    """deeplio_nets.py:8-99"""

    def __init__(self, input_shape, cfg, bn_d=0.1):
        super().__init__()
        cc = get_config_container()
        self.cfg_container = cc
        self.seq_size, self.combinations = cc.seq_size, cc.combinations
        self.cfg = cfg['deeplio']
        self.p = self.cfg.get('dropout', 0.)
        self.input_shape = input_shape
        self.lidar_feat_net = self.imu_feat_net = self.fusion_net = self.odom_feat_net = None
        self.drop = self.fc_pos = self.fc_ori = None
        self.side_stream = True      # overlap the IMU branch with the lidar branch
        self._side = None
        # data parallel: called (from backward) when the gradients of everything behind the
        # feature nets -- odometry net, heads -- are complete (deeplio_amd.dist.GradSync)
        self.tail_grads_ready = None

    def initialize(self):
        last = next(n for n in (self.odom_feat_net, self.fusion_net, self.imu_feat_net,
                                self.lidar_feat_net) if n is not None)
        in_shape = last.get_output_shape()[2]
        if self.p > 0:
            self.drop = nn.Dropout(self.p)      # container for parity of module lists; not called
        self.fc_pos = nn.Linear(in_shape, 3)
        self.fc_ori = nn.Linear(in_shape, 3)

    def forward(self, x):
        return self.forward_tail(self.forward_features(x))

    def forward_features(self, x, defer_imu=True):
        """The wide part of the step: IMU net and the two lidar encoders, on their own streams.
        -> dict(lidar=(fa, fb, (b, s)) | None, imu=feature | None, imu_stream=stream | None); with defer_imu
        the IMU feature is re-attached through DeferredBranchFn (its backward then runs where autograd
        reaches it), else it is returned on its own tape and the caller runs its backward."""
        lidar_imgs, imu_meas = x[0], x[1]
        enc = feat_i = None
        if self.training and getattr(self, "_bn_counters", None) is not None:
            self._bn_counters += 1
        # The IMU branch (latency-bound persistent RNN kernels, 2 workgroups) is independent of the
        # lidar branch until the fusion layer: run it on a second HIP stream so it overlaps the
        # convolutions; autograd replays each backward node on its forward stream.
        side = None
        if (self.lidar_feat_net is not None and self.imu_feat_net is not None and torch.is_tensor(imu_meas)
                and imu_meas.is_cuda and self.side_stream):
            if self._side is None:
                self._side = Fh.aux_stream(imu_meas.device, "imu")
            side = self._side
            main = torch.cuda.current_stream()
            side.wait_stream(main)
            with Fh.on_stream(side):
                feat_i = self.imu_feat_net(imu_meas)
        elif self.imu_feat_net is not None:     # same issue order (dropout counter) as the overlapped path
            feat_i = self.imu_feat_net(imu_meas)
        if self.lidar_feat_net is not None:
            enc = self.lidar_feat_net.encode(lidar_imgs)
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
            feat_i.record_stream(torch.cuda.current_stream())
            if defer_imu and _DEFER_IMU and torch.is_grad_enabled() and feat_i.requires_grad:
                feat_i = Fh.DeferredBranchFn.attach(feat_i, side)
        return {"lidar": enc, "imu": feat_i, "imu_stream": side}

    def forward_tail(self, feats, grads_ready_hook=True):
        """Everything behind the feature extractors: lidar fusion + fc1, fusion net, odometry net, heads --
        ~100 small dependent launches."""
        last = feat_l = None
        feat_i = feats["imu"]
        if feats["lidar"] is not None:
            last = feat_l = self.lidar_feat_net.head(*feats["lidar"])
        if feat_i is not None:
            last = feat_i
        if self.fusion_net is not None:
            last = self.fusion_net([feat_l, feat_i])
        if (grads_ready_hook and self.tail_grads_ready is not None and self.training and torch.is_tensor(last)
                and last.requires_grad):
            cb = self.tail_grads_ready
            last.register_hook(lambda g: cb())
        K = self.fc_pos.in_features
        if (Fh._TAIL_FUSED[0] and isinstance(self.odom_feat_net, OdomFeatRNN) and torch.is_tensor(last) and last.is_cuda
                and last.dim() == 3 and self.odom_feat_net.hidden_size == K and K % 4 == 0 and K <= 8192):
            # the heads read the forward half of the LSTM output in place: no slice copy, dropout + both heads one launch
            full = self.odom_feat_net.forward_full(last)
            return Fh.HeadsFn.apply(full, self.fc_pos.weight, self.fc_pos.bias, self.fc_ori.weight, self.fc_ori.bias,
                                    self.p, self.training)
        if self.odom_feat_net is not None:
            last = self.odom_feat_net(last)
        if (Fh._TAIL_FUSED[0] and torch.is_tensor(last) and last.is_cuda and last.dim() == 3 and last.shape[-1] == K
                and last.dtype == torch.float32 and ops.heads_ok(last.shape[0] * last.shape[1], K, K)):
            return Fh.HeadsFn.apply(last, self.fc_pos.weight, self.fc_pos.bias, self.fc_ori.weight, self.fc_ori.bias,
                                    self.p, self.training)
        last = Fh.dropout(last, self.p, self.training)
        x_pos = Fh.LinearFn.apply(last, self.fc_pos.weight, self.fc_pos.bias, ops.ACT_NONE)
        x_ori = Fh.LinearFn.apply(last, self.fc_ori.weight, self.fc_ori.bias, ops.ACT_NONE)
        return x_pos, x_ori

    def get_feat_networks(self):
        nets = []
        for n in (self.odom_feat_net, self.fusion_net, self.imu_feat_net, self.lidar_feat_net):
            if n is not None and isinstance(n, nn.Module):
                nets.extend(n.get_modules())
        return nets


# ------------------------------------------------------------------------------ factory
class _Log:
    def info(self, *a, **k):
        pass

    print = error = warning = info


net_logger = _Log()

LIDAR_NETS = {'lidar-feat-pointseg': LidarPointSegFeat, 'lidar-feat-flownet': LidarFlowNetFeat,
              'lidar-feat-resnet': LidarResNetFeat, 'lidar-feat-simple-1': LidarSimpleFeat1}
IMU_NETS = {'imu-feat-fc': ImuFeatFC, 'imu-feat-rnn': ImufeatRNN0}
FUSION_NETS = {'fusion-layer-cat': DeepLIOFusionCat, 'fusion-layer-soft': DeepLIOFusionSoft}
ODOM_NETS = {'odom-feat-fc': OdomFeatFC, 'odom-feat-rnn': OdomFeatRNN}


def get_model(input_shape, cfg, device):
    """nets/__init__.py:16-20"""
    return create_deeplio_arch(input_shape, cfg, device)


def load_state_dict(module, model_path):
    """nets/__init__.py:216-223: checkpoints are {'state_dict': ...} .tar files"""
    state = torch.load(model_path, map_location=module.device)
    module.load_state_dict(state['state_dict'])


def disable_grad(module):
    for p in module.parameters():
        p.requires_grad = False


def _select(table, name, err):
    if name is None:
        return None, None
    key = name.lower()
    if key not in table:
        raise ValueError(err.format(key))
    return table[key], key


def _finish(net, feat_cfg, device, pointseg=False):
    net.to(device)
    if feat_cfg.get('pretrained', False):
        path = feat_cfg['model-path']
        if pointseg and 'encoder' in path:
            load_state_dict(net.encoder1, path)      # nets/__init__.py:110-112
        else:
            load_state_dict(net, path)
        net.pretrained = True
    if not feat_cfg.get('requires-grad', True):
        disable_grad(net)
    return net


def create_deeplio_arch(input_shape, cfg, device):
    """nets/__init__.py:23-78"""
    arch = cfg['deeplio']
    net = DeepLIO(input_shape, cfg)
    lidar_shape = imu_shape = fusion_shape = None

    kls, key = _select(LIDAR_NETS, arch['lidar-feat-net'].get('name', None), "Wrong feature network {}")
    if kls is not None:
        net.lidar_feat_net = _finish(kls(input_shape, cfg[key]), arch['lidar-feat-net'], device,
                                     pointseg=key == 'lidar-feat-pointseg')
        lidar_shape = net.lidar_feat_net.get_output_shape()
    kls, key = _select(IMU_NETS, arch['imu-feat-net'].get('name', None), "Wrong feature network {}")
    if kls is not None:
        net.imu_feat_net = _finish(kls(cfg[key]), arch['imu-feat-net'], device)
        imu_shape = net.imu_feat_net.get_output_shape()
    if lidar_shape is not None and imu_shape is not None:
        kls, key = _select(FUSION_NETS, (arch.get('fusion-net') or {}).get('name', None),
                           "Wrong feature network {}")
        if kls is not None:
            net.fusion_net = kls([lidar_shape, imu_shape], cfg[key])
            fusion_shape = net.fusion_net.get_output_shape()

    if fusion_shape is not None:
        odom_in = fusion_shape
    elif lidar_shape is not None:
        odom_in = lidar_shape
    elif imu_shape is not None:
        odom_in = imu_shape
    else:
        raise ValueError("No input-shape for odometry network is defined, please check you configuration!")

    kls, key = _select(ODOM_NETS, arch['odom-feat-net'].get('name', None), "Wrong odometry feature network {}")
    if kls is not None:
        net.odom_feat_net = _finish(kls(odom_in[2], cfg[key]), arch['odom-feat-net'], device)

    net.initialize()
    net.to(device=device)
    share_bn_counters(net)
    if arch.get('pretrained', False):
        load_state_dict(net, arch['model-path'])
        net.pretrained = True
    return net
