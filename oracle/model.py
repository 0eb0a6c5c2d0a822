"""CPU restatement of the DeepLIO model family, losses, optimizer factory and LR schedule.

TEST INFRASTRUCTURE (see oracle/__init__.py): the checker for the HIP path and the timed
CPU baseline ("port") of bench.py.  Written from the behaviour of the reference, with the
same state_dict keys so reference-style weights load by key; every class cites the
reference site it follows.  Pinned against goldens captured from the reference itself
(tests/golden/).

Deliberate, documented differences from the reference:
  * shapes are derived arithmetically instead of running dummy forwards
    (lidar_feat_nets.py:33-40, pointseg_net.py:73-79), so construction leaves the module in
    train mode instead of eval mode -- callers set the mode explicitly;
  * lidar `fusion: cat` builds fc1 with 2F inputs (the reference crashes, Q1);
  * DeepLIOFusionSoft multiplies out of place (same forward values; the reference's in-place
    `*=` breaks autograd for some configurations, Q2).
"""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import se3


# ----------------------------------------------------------------------------------------------
class Ctx:
    """What the reference keeps in its global ConfigContainer (models/misc.py:167-195)."""

    def __init__(self, cfg):
        comb = cfg['datasets']['combinations']
        self.combinations = comb
        self.seq_size = len(comb)
        self.timestamps = len(comb[0])


def _named(mod):
    return type(mod).__name__.lower()


# ------------------------------------------------------------------------------ PointSeg encoder
class Fire(nn.Module):
    """pointseg_modules.py:86-142: squeeze 1x1 -> BN -> ReLU -> {expand1x1, expand3x3} -> BN ->
    ReLU -> cat; 'simple' bypass adds the input when in == out planes."""

    def __init__(self, cin, sq, e1, e3, bn_d=0.1, bypass=None):
        super().__init__()
        self.squeeze = nn.Conv2d(cin, sq, 1)
        self.squeeze_bn = nn.BatchNorm2d(sq, momentum=bn_d)
        self.expand1x1 = nn.Conv2d(sq, e1, 1)
        self.expand1x1_bn = nn.BatchNorm2d(e1, momentum=bn_d)
        self.expand3x3 = nn.Conv2d(sq, e3, 3, padding=1)
        self.expand3x3_bn = nn.BatchNorm2d(e3, momentum=bn_d)
        self.residual = bypass == "simple" and cin == e1 + e3
        if bypass == "complex" and cin != e1 + e3:
            self.upsample = nn.Conv2d(cin, e1 + e3, 1)

    def forward(self, x):
        s = F.relu(self.squeeze_bn(self.squeeze(x)))
        a = F.relu(self.expand1x1_bn(self.expand1x1(s)))
        b = F.relu(self.expand3x3_bn(self.expand3x3(s)))
        out = torch.cat([a, b], 1)
        if hasattr(self, "upsample"):
            out = out + self.upsample(x)
        elif self.residual:
            out = out + x
        return out


class SELayer(nn.Module):
    """pointseg_modules.py:203-221."""

    def __init__(self, c, reduction):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(c, c // reduction, bias=False), nn.ReLU(),
                                nn.Linear(c // reduction, c, bias=False), nn.Sigmoid())

    def forward(self, x):
        y = self.fc(x.mean((2, 3)))
        return x * y[:, :, None, None]


PS_BLOCKS = (  # pointseg_net.py:24-55: (fires [(cin, sq, e)], se channels or None, pool stride or None)
    ("fire_blk1", [(64, 16, 64), (128, 16, 64)], 128, (1, 2)),
    ("fire_blk2", [(128, 32, 128), (256, 32, 128)], 256, (1, 2)),
    ("fire_blk3", [(256, 48, 192), (384, 48, 192), (384, 64, 256), (512, 64, 256)], 512, (2, 2)),
    ("fire_blk4", [(512, 64, 256), (512, 64, 256)], 512, (2, 2)),
    ("fire_blk5", [(512, 80, 384), (768, 80, 384)], None, None),
)


class PSEncoder(nn.Module):
    """pointseg_net.py:9-71."""

    def __init__(self, cin, bypass, bn_d=0.1):
        super().__init__()
        self.conv1a = nn.Sequential(nn.Conv2d(cin, 64, (3, 5), (1, 2), (1, 2)),
                                    nn.BatchNorm2d(64, momentum=bn_d), nn.ReLU())
        self.pool1 = nn.MaxPool2d(3, (1, 2), 1)
        for name, fires, se, pool in PS_BLOCKS:
            mods = []
            for i, (ci, sq, e) in enumerate(fires):
                last_of_net = name == "fire_blk5" and i == len(fires) - 1
                mods.append(Fire(ci, sq, e, e, bn_d, None if last_of_net else bypass))
            if se:
                mods.append(SELayer(se, 2))
            if pool:
                mods.append(nn.MaxPool2d(3, pool, 1))
            setattr(self, name, nn.Sequential(*mods))
        self.out_channels = 768

    def forward(self, x):
        x = self.pool1(self.conv1a(x))
        for name, *_ in PS_BLOCKS:
            x = getattr(self, name)(x)
        return x


# ------------------------------------------------------------------------------ FlowNet encoder
def _cbr(cin, cout, k=(3, 3), stride=1):
    """base_net.py:55-71 with batch_norm=True."""
    return nn.Sequential(nn.Conv2d(cin, cout, k, stride, ((k[0] - 1) // 2, (k[1] - 1) // 2), bias=False),
                         nn.BatchNorm2d(cout), nn.ReLU())


FLOWNET_LAYERS = (  # lidar_feat_nets.py:248-257
    ("conv1", None, 64, (5, 7), (1, 2)), ("conv2", 64, 128, (3, 5), (1, 2)),
    ("conv3", 128, 256, (3, 5), (1, 2)), ("conv3_1", 256, 256, (3, 3), 1),
    ("conv4", 256, 512, (3, 3), 2), ("conv4_1", 512, 512, (3, 3), 1),
    ("conv5", 512, 512, (3, 3), 2), ("conv5_1", 512, 512, (3, 3), 1),
    ("conv6", 512, 1024, (3, 3), 2),
)


class FlowNetEncoder(nn.Module):
    """lidar_feat_nets.py:240-267."""

    def __init__(self, cin):
        super().__init__()
        for name, ci, co, k, s in FLOWNET_LAYERS:
            setattr(self, name, _cbr(cin if ci is None else ci, co, k, s))
        self.out_channels = 1024

    def forward(self, x):
        for name, *_ in FLOWNET_LAYERS:
            x = getattr(self, name)(x)
        return x.mean((2, 3))


# ------------------------------------------------------------------------------ ResNet encoder
class BasicBlock(nn.Module):
    """torchvision.models.resnet.BasicBlock (external to the reference, version unpinned;
    semantics stable since torchvision 0.5): conv3x3(stride)-BN-ReLU-conv3x3-BN,
    (+ downsample(x) | x), ReLU."""
    expansion = 1

    def __init__(self, cin, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = F.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return F.relu(out + idt)


class ResNetEncoder(nn.Module):
    """resnet.py:14-112 with layers [3,3,3,2]."""
    STAGES = ((64, 3, (1, 2)), (128, 3, (1, 2)), (256, 3, (2, 2)), (512, 2, (2, 2)))

    def __init__(self, cin):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, 64, (5, 7), 1, (2, 3), bias=True)
        self.bn1 = nn.BatchNorm2d(64)
        self.maxpool = nn.MaxPool2d(3, (1, 2), 1)
        inpl = 64
        for i, (planes, nblk, stride) in enumerate(self.STAGES):
            down = nn.Sequential(nn.Conv2d(inpl, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
            blocks = [BasicBlock(inpl, planes, stride, down)]
            blocks += [BasicBlock(planes, planes) for _ in range(nblk - 1)]
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*blocks))
            inpl = planes
        for m in self.modules():   # resnet.py:51-56
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1.)
                nn.init.constant_(m.bias, 0.)
        self.out_channels = 512

    def forward(self, x):
        x = self.maxpool(F.relu(self.bn1(self.conv1(x))))
        for i in range(4):
            x = getattr(self, "layer%d" % (i + 1))(x)
        return x.mean((2, 3))


# ------------------------------------------------------------------------------ Simple-1 encoder
SIMPLE_LAYERS = (  # lidar_feat_nets.py:279-304: (idx, cin, cout, k, stride, pad, pool stride | None)
    (1, None, 64, (5, 7), (1, 2), (2, 3), (1, 2)), (2, 64, 128, (3, 5), 1, (1, 2), (1, 2)),
    (3, 128, 128, 3, 1, 1, None), (4, 128, 256, 3, 1, 1, (2, 2)),
    (5, 256, 256, 3, 1, 1, None), (6, 256, 512, 3, 1, 1, (2, 2)),
    (7, 512, 512, 3, 1, 1, None),
)


class FeatureNetSimple1(nn.Module):
    """lidar_feat_nets.py:270-342: conv(bias) -> ReLU -> BN, ceil-mode pools.  bypass=True is
    undefined in the reference (crashes, Q7) and is rejected here."""

    def __init__(self, cin, bypass=False):
        super().__init__()
        if bypass:
            raise ValueError("lidar-feat-simple-1 bypass=true has no reference behaviour")
        for i, ci, co, k, s, p, pool in SIMPLE_LAYERS:
            setattr(self, "conv%d" % i, nn.Conv2d(cin if ci is None else ci, co, k, s, p))
            setattr(self, "bn%d" % i, nn.BatchNorm2d(co))
        self.out_channels = 512

    def forward(self, x):
        for i, _, _, _, _, _, pool in SIMPLE_LAYERS:
            x = getattr(self, "bn%d" % i)(F.relu(getattr(self, "conv%d" % i)(x)))
            if pool:
                x = F.max_pool2d(x, 3, pool, 1, ceil_mode=True)
        return x.mean((2, 3))


# ------------------------------------------------------------------------------ lidar feature nets
class _LidarFeat(nn.Module):
    """Two-stream wrapper (lidar_feat_nets.py:46-237): pair stacked on channels per stream,
    encoder1(xyz) / encoder2(normals), GAP, add|sub|cat, fc1 (+act) and dropout."""
    enc = None
    feat = None
    act = staticmethod(F.relu)
    drop_before_fc = False

    def __init__(self, input_shape, cfg, ctx):
        super().__init__()
        c, h, w = input_shape
        self.p = cfg['dropout']
        self.fusion = cfg['fusion']
        self.encoder1 = self.make_encoder(2 * c, cfg)
        self.encoder2 = self.make_encoder(2 * c, cfg)
        if self.p > 0.:
            self.drop = nn.Dropout(self.p)
        nfeat = self.encoder1.out_channels * (2 if self.fusion == 'cat' else 1)
        self.fc1 = nn.Linear(nfeat, 128)
        self.output_shape = torch.Size([1, ctx.seq_size, 128])
        self.pretrained = False

    name = property(_named)

    def get_output_shape(self):
        return self.output_shape

    def get_modules(self):
        return [self]

    def streams(self, a, b):
        fa, fb = self.encoder1(a), self.encoder2(b)
        if fa.dim() == 4:
            fa, fb = fa.mean((2, 3)), fb.mean((2, 3))
        if self.fusion == 'cat':
            return torch.cat((fa, fb), 1)
        return fa + fb if self.fusion == 'add' else fa - fb

    def forward(self, x):
        xyz, nrm = x
        b, s, t, c, h, w = xyz.shape
        y = self.streams(xyz.reshape(b * s, t * c, h, w), nrm.reshape(b * s, t * c, h, w))
        if self.drop_before_fc:
            if self.p > 0.:
                y = self.drop(y)
            y = self.act(self.fc1(y))
        else:
            y = self.act(self.fc1(y))
            if self.p > 0.:
                y = self.drop(y)
        return y.view(b, s, -1)


class LidarPointSegFeat(_LidarFeat):     # lidar_feat_nets.py:46-101
    def make_encoder(self, cin, cfg):
        return PSEncoder(cin, cfg['bypass'])


class LidarFlowNetFeat(_LidarFeat):      # lidar_feat_nets.py:104-148
    def make_encoder(self, cin, cfg):
        return FlowNetEncoder(cin)


class LidarResNetFeat(_LidarFeat):       # lidar_feat_nets.py:151-189 (dropout BEFORE fc1)
    drop_before_fc = True

    def make_encoder(self, cin, cfg):
        return ResNetEncoder(cin)


class LidarSimpleFeat1(_LidarFeat):      # lidar_feat_nets.py:192-237 (dropout, fc1, leaky-ReLU)
    drop_before_fc = True
    act = staticmethod(F.leaky_relu)

    def make_encoder(self, cin, cfg):
        return FeatureNetSimple1(cin, cfg['bypass'])


# ------------------------------------------------------------------------------ IMU / fusion / odom
class _Feat(nn.Module):
    name = property(_named)

    def __init__(self):
        super().__init__()
        self.pretrained = False

    def get_output_shape(self):
        return self.output_shape

    def get_modules(self):
        return [self]


class ImuFeatFC(_Feat):
    """imu_feat_nets.py:21-53: per-sample MLP with leaky-ReLU, then SUM over the T samples."""

    def __init__(self, cfg, ctx):
        super().__init__()
        hs = cfg.get('hidden-size', [6, 6])
        self.p = cfg['dropout']
        dims = [cfg['input-size']] + list(hs)
        self.net = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        if self.p > 0.:
            self.dropout = nn.Dropout(self.p)
        self.output_shape = [1, ctx.seq_size, hs[-1]]

    def forward(self, x):
        # per-(b, s) loops as in the reference: a batched restatement is mathematically equal
        # but rounds differently, and Adam amplifies that to 1e-3 within five steps
        outs = []
        nb, ns = len(x), len(x[0])
        for b in range(nb):
            for s in range(ns):
                y = x[b][s]
                for m in self.net:
                    y = F.leaky_relu(m(y), 0.01)
                if self.p > 0.:
                    y = self.dropout(y)
                outs.append(torch.sum(y, dim=0))
        return torch.stack(outs).view(nb, ns, -1)


def _make_rnn(cfg, input_size, p):
    kind = nn.GRU if cfg['type'].lower() == 'gru' else nn.LSTM
    return kind(input_size=input_size, hidden_size=cfg.get('hidden-size', 6),
                num_layers=cfg.get('num-layers', 2), bidirectional=cfg.get('bidirectional', False),
                dropout=p, batch_first=True)


class ImufeatRNN0(_Feat):
    """imu_feat_nets.py:56-83: the RNN state is CARRIED from sub-sequence s to s+1; the
    feature is the last time step of the forward direction of the top layer."""

    def __init__(self, cfg, ctx):
        super().__init__()
        self.hidden_size = cfg.get('hidden-size', 6)
        self.num_dir = 2 if cfg.get('bidirectional', False) else 1
        self.rnn = _make_rnn(cfg, cfg['input-size'], cfg['dropout'])
        self.output_shape = [1, ctx.seq_size, self.hidden_size]

    def forward(self, x):
        b, s, t, n = x.shape
        state, outs = None, []
        for i in range(s):
            out, state = self.rnn(x[:, i], state)
            outs.append(out.view(b, t, self.num_dir, self.hidden_size)[:, -1, 0, :])
        return torch.stack(outs, 1)


class DeepLIOFusionCat:
    """fusion_nets.py:9-37 (a plain class in the reference: no parameters, not in state_dict)."""
    name = "deepliofusioncat"

    def __init__(self, in_shapes, cfg, ctx):
        self.output_shape = [1, ctx.seq_size, sum(s[-1] for s in in_shapes)]

    def get_output_shape(self):
        return self.output_shape

    def __call__(self, x):
        return torch.cat((x[0], x[1]), 2)


class DeepLIOFusionSoft(_Feat):
    """fusion_nets.py:40-78: sigmoid gates from the concatenated features re-weight each
    modality (out of place here, Q2)."""

    def __init__(self, in_shapes, cfg, ctx):
        super().__init__()
        total = sum(s[-1] for s in in_shapes)
        self.layers = nn.ModuleList([nn.Linear(total, s[-1]) for s in in_shapes])
        self.output_shape = [1, ctx.seq_size, total]

    def forward(self, x):
        lidar, imu = x
        cat = torch.cat((lidar, imu), 2)
        s1, s2 = torch.sigmoid(self.layers[0](cat)), torch.sigmoid(self.layers[1](cat))
        return torch.cat((lidar * s1, imu * s2), 2)


class OdomFeatFC(_Feat):
    """odom_feat_nets.py:8-45 (reads key 'hidden-size'; config.yaml gives 'size' -> default
    [256,128], Q4)."""

    def __init__(self, in_features, cfg, ctx):
        super().__init__()
        hs = cfg.get('hidden-size', [256, 128])
        self.p = cfg.get('dropout', 0.)
        dims = [in_features] + list(hs)
        self.layers = nn.ModuleList([nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])])
        if self.p > 0.:
            self.drop = nn.Dropout(self.p)
        self.output_shape = [1, 1, hs[-1]]

    def forward(self, x):
        for layer in self.layers:
            x = F.leaky_relu(layer(x), 0.01)
        if self.p > 0.:
            x = self.drop(x)
        return x


class OdomFeatRNN(_Feat):
    """odom_feat_nets.py:48-86: RNN over the S axis, forward-direction half kept."""

    def __init__(self, in_features, cfg, ctx):
        super().__init__()
        self.hidden_size = cfg.get('hidden-size', 6)
        self.num_dir = 2 if cfg.get('bidirectional', False) else 1
        self.rnn = _make_rnn(cfg, in_features, cfg.get('dropout', 0.))
        self.output_shape = [1, 1, self.hidden_size]

    def forward(self, x):
        b, s, _ = x.shape
        out, _ = self.rnn(x)
        return out.view(b, s, self.num_dir, self.hidden_size)[:, :, 0]


# ------------------------------------------------------------------------------ top module + factory
class DeepLIO(_Feat):
    """deeplio_nets.py:27-99."""

    def __init__(self, input_shape, cfg):
        super().__init__()
        self.p = cfg['deeplio'].get('dropout', 0.)
        self.input_shape = input_shape
        self.lidar_feat_net = self.imu_feat_net = self.fusion_net = self.odom_feat_net = None
        self.drop = self.fc_pos = self.fc_ori = None

    def initialize(self):
        last = next(n for n in (self.odom_feat_net, self.fusion_net, self.imu_feat_net,
                                self.lidar_feat_net) if n is not None)
        nin = last.get_output_shape()[2]
        if self.p > 0:
            self.drop = nn.Dropout(self.p)
        self.fc_pos = nn.Linear(nin, 3)
        self.fc_ori = nn.Linear(nin, 3)

    def forward(self, x):
        imgs, imus = x
        feat = fl = fi = None
        if self.lidar_feat_net is not None:
            feat = fl = self.lidar_feat_net(imgs)
        if self.imu_feat_net is not None:
            feat = fi = self.imu_feat_net(imus)
        if self.fusion_net is not None:
            feat = self.fusion_net([fl, fi])
        if self.odom_feat_net is not None:
            feat = self.odom_feat_net(feat)
        if self.p > 0.:
            feat = self.drop(feat)
        return self.fc_pos(feat), self.fc_ori(feat)

    def get_feat_networks(self):
        nets = []
        for n in (self.odom_feat_net, self.fusion_net, self.imu_feat_net, self.lidar_feat_net):
            if n is not None and isinstance(n, nn.Module):
                nets.extend(n.get_modules())
        return nets


LIDAR_NETS = {'lidar-feat-pointseg': LidarPointSegFeat, 'lidar-feat-flownet': LidarFlowNetFeat,
              'lidar-feat-resnet': LidarResNetFeat, 'lidar-feat-simple-1': LidarSimpleFeat1}
IMU_NETS = {'imu-feat-fc': ImuFeatFC, 'imu-feat-rnn': ImufeatRNN0}
FUSION_NETS = {'fusion-layer-cat': DeepLIOFusionCat, 'fusion-layer-soft': DeepLIOFusionSoft}
ODOM_NETS = {'odom-feat-fc': OdomFeatFC, 'odom-feat-rnn': OdomFeatRNN}


def _pick(table, name, what):
    if name is None:
        return None
    try:
        return table[name.lower()]
    except KeyError:
        raise ValueError("Wrong %s network %s" % (what, name))


def get_model(input_shape, cfg, device="cpu"):
    """nets/__init__.py:16-78 (construction + wiring; checkpoint loading left to callers)."""
    ctx = Ctx(cfg)
    arch = cfg['deeplio']
    net = DeepLIO(input_shape, cfg)
    shapes = {}
    kls = _pick(LIDAR_NETS, arch['lidar-feat-net'].get('name'), 'feature')
    if kls:
        net.lidar_feat_net = kls(input_shape, cfg[arch['lidar-feat-net']['name'].lower()], ctx)
        shapes['lidar'] = net.lidar_feat_net.get_output_shape()
    kls = _pick(IMU_NETS, arch['imu-feat-net'].get('name'), 'feature')
    if kls:
        net.imu_feat_net = kls(cfg[arch['imu-feat-net']['name'].lower()], ctx)
        shapes['imu'] = net.imu_feat_net.get_output_shape()
    odom_in = shapes.get('imu', shapes.get('lidar'))
    if 'lidar' in shapes and 'imu' in shapes:
        fname = (arch.get('fusion-net') or {}).get('name')
        kls = _pick(FUSION_NETS, fname, 'feature')
        if kls:
            net.fusion_net = kls([shapes['lidar'], shapes['imu']], cfg[fname.lower()], ctx)
            odom_in = net.fusion_net.get_output_shape()
        else:
            odom_in = shapes['lidar']
    elif 'lidar' in shapes:
        odom_in = shapes['lidar']
    if odom_in is None:
        raise ValueError("No input-shape for odometry network is defined, please check you configuration!")
    kls = _pick(ODOM_NETS, arch['odom-feat-net'].get('name'), 'odometry feature')
    if kls:
        net.odom_feat_net = kls(odom_in[2], cfg[arch['odom-feat-net']['name'].lower()], ctx)
    net.initialize()
    for key, sub in (('lidar-feat-net', net.lidar_feat_net), ('imu-feat-net', net.imu_feat_net),
                     ('odom-feat-net', net.odom_feat_net)):
        if sub is not None and not arch[key].get('requires-grad', True):
            for p in sub.parameters():
                p.requires_grad = False
    return net.to(device)


# ------------------------------------------------------------------------------ losses / optimizer
def geodesic_theta2(a, b):
    """Squared rotation angle between the rotations two quaternions [..., 4] stand for:
    theta = 2 atan2(|a ^ b|, |<a, b>|) (= 2 acos|<a, b>| for unit quaternions = |log(Ra^T Rb)|),
    invariant to sign, scale and component order.  No reference counterpart (the reference's
    rotation terms are MSE, losses/losses.py:71-85): this restates the definition in
    include/deeplio_hip.h (dlio_pose_loss_fwd, mode bit 1) for BASELINE configs[4]."""
    d = (a * b).sum(-1)
    n2 = 0.
    for i in range(4):
        for j in range(i + 1, 4):
            n2 = n2 + (a[..., i] * b[..., j] - a[..., j] * b[..., i]) ** 2
    # sqrt'(0) is infinite: keep autograd finite at identical rotations (the limit gradient is 0)
    n = torch.sqrt(n2.clamp_min(1e-300 if a.dtype == torch.float64 else 1e-36))
    return (2. * torch.atan2(n, d.abs())) ** 2


def so3_to_quat(phi):
    """unit quaternion (w, x, y, z) of exp(phi^), phi [..., 3]"""
    t = phi.norm(dim=-1, keepdim=True)
    small = t < 1e-3
    ts = torch.where(small, torch.ones_like(t), t)
    k = torch.where(small, 0.5 - t * t / 48., torch.sin(0.5 * ts) / ts)
    return torch.cat([torch.cos(0.5 * t), k * phi], -1)


def _rot_terms(rotation, pw, gw, pq, gq, use_local, use_global):
    if rotation == 'geodesic':
        Lw = geodesic_theta2(so3_to_quat(pw), so3_to_quat(gw)).mean() if use_local else 0.
        Lq = geodesic_theta2(pq, gq).mean() if use_global else 0.
    else:
        Lw = F.mse_loss(pw, gw) if use_local else 0.
        Lq = F.mse_loss(pq, gq) if use_global else 0.
    return Lw, Lq


class HWSLoss(nn.Module):
    """losses/losses.py:51-86."""
    rotation = 'mse'

    def __init__(self, sx=0., sq=-2.5, learn=True, loss_Types=(True, True)):
        super().__init__()
        self.loss_Types = list(loss_Types)
        self.sx = nn.Parameter(torch.tensor(float(sx)), requires_grad=learn)
        self.sq = nn.Parameter(torch.tensor(float(sq)), requires_grad=learn)

    def forward(self, pt, pw, pp, pq, gt, gw, gp, gq):
        Lt = F.mse_loss(pt, gt) if self.loss_Types[0] else 0.
        Lp = F.mse_loss(pp, gp) if self.loss_Types[1] else 0.
        Lw, Lq = _rot_terms(self.rotation, pw, gw, pq, gq, self.loss_Types[0], self.loss_Types[1])
        return (Lp + Lt) * torch.exp(-self.sx) + self.sx + (Lq + Lw) * torch.exp(-self.sq) + self.sq


class LWSLoss(nn.Module):
    """losses/losses.py:11-39."""
    rotation = 'mse'

    def __init__(self, beta=1125., loss_Types=(True, True)):
        super().__init__()
        self.beta = beta
        self.loss_Types = list(loss_Types)

    def forward(self, pt, pw, pp, pq, gt, gw, gp, gq):
        Lt = F.mse_loss(pt, gt) if self.loss_Types[0] else 0.
        Lp = F.mse_loss(pp, gp) if self.loss_Types[1] else 0.
        Lw, Lq = _rot_terms(self.rotation, pw, gw, pq, gq, self.loss_Types[0], self.loss_Types[1])
        return (Lp + Lt) + self.beta * (Lq + Lw)


def get_loss_function(cfg, device="cpu"):
    """losses/__init__.py:4-30."""
    lc = cfg['losses']
    name = lc['active'].lower()
    params = lc.get(name, {}).get('params', {})
    lt = lc['loss-type'].lower()
    if "+" in lt:
        types = [True, True]
    elif lt == "global":
        types = [False, True]
    elif lt == "local":
        types = [True, False]
    else:
        raise ValueError("Wrong loss type selected!")
    if name == 'hwsloss':
        loss = HWSLoss(params.get('sx', 0.), params.get('sq', -2.5), params.get('learn', False), types).to(device)
    elif name == 'lwsloss':
        loss = LWSLoss(params.get('beta', 1125.), types).to(device)
    else:
        raise ValueError("Loss {} is not supported!".format(name))
    loss.rotation = str(lc.get('rotation', 'mse')).lower()      # 'geodesic': BASELINE configs[4]
    return loss


def create_optimizer(params, cfg, lr, weight_decay, momentum=0.9):
    """models/optimizer.py:4-16."""
    kind = cfg['optimizer'].lower()
    if kind == 'sgd':
        return torch.optim.SGD(params, lr=lr, weight_decay=weight_decay, momentum=momentum)
    if kind == 'adam':
        return torch.optim.Adam(params, lr=lr, weight_decay=weight_decay)
    if kind == 'rmsprop':
        return torch.optim.RMSprop(params, lr=lr, weight_decay=weight_decay)
    if kind == 'adadelta':
        return torch.optim.Adadelta(params, lr=lr, weight_decay=weight_decay)
    raise ValueError("Optimizer {} not supported!".format(kind))


def poly_lr(base_lr, epoch, max_decay_steps, end_lr=1e-6, power=2.0):
    """PolynomialLRDecay.get_lr (models/misc.py:147-156)."""
    if epoch > max_decay_steps:
        return end_lr
    return (base_lr - end_lr) * ((1 - epoch / max_decay_steps) ** power) + end_lr


def train_step(model, criterion, optimizer, batch, max_glob_seq=2):
    """One iteration of Trainer.train (trainer.py:213-281) without logging / NaN guards."""
    xyz, nrm, imu, gt_f2f, gt_f2g = batch
    pt, pw = model([[xyz, nrm], imu])
    pp, pq = se3.se3_to_SE3(pt, pw)
    lt = criterion.loss_Types
    if lt[0] and not lt[1]:
        pp, pq = pp.detach(), pq.detach()
    elif lt[1] and not lt[0]:
        pt, pw = pt.detach(), pw.detach()
    sl = slice(1, max_glob_seq + 1)
    loss = criterion(pt, pw, pp[:, sl], pq[:, sl], gt_f2f[:, :, 0:3], gt_f2f[:, :, 3:],
                     gt_f2g[:, sl, 0:3], gt_f2g[:, sl, 3:7])
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return loss.detach()
