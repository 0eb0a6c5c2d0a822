"""Mirror of deeplio/models/misc.py: the module-global ConfigContainer (misc.py:167-195) that
every net constructor reads, and PolynomialLRDecay (misc.py:131-164)."""
import numpy as np


class ConfigContainer:
    def __init__(self, cfg, args):
        self.cfg = cfg
        self.args = args
        self.ds_cfg = cfg['datasets']
        self.curr_dataset_cfg = cfg['datasets'][cfg['current-dataset']]
        self.combinations = np.array(self.ds_cfg['combinations'])
        self.seq_size = len(self.combinations)
        self.timestamps = len(self.combinations[0])
        self.device = args.device
        self.batch_size = args.batch_size
        self.seq_size_data = self.ds_cfg['sequence-size']


config_container = None


def get_config_container():
    if config_container is None:
        raise ValueError("Config container must be created by Worker first!")
    return config_container


def build_config_container(cfg, args):
    global config_container
    config_container = ConfigContainer(cfg, args)
    return config_container


class PolynomialLRDecay:
    """lr(e) = (lr0 - end) * (1 - e/E)^power + end for e <= E, else end; stepped per epoch
    (trainer.py:113-114,172).  Works with any optimizer exposing `param_groups`."""

    def __init__(self, optimizer, max_decay_steps, end_learning_rate=0.0001, power=1.0, last_epoch=-1):
        if max_decay_steps <= 1.:
            raise ValueError('max_decay_steps should be greater than 1.')
        self.optimizer = optimizer
        self.max_decay_steps = max_decay_steps
        self.end_learning_rate = end_learning_rate
        self.power = power
        # torch.optim.lr_scheduler._LRScheduler's protocol (the reference's class derives from it, misc.py:131):
        # a fresh schedule records every group's starting rate as 'initial_lr', a resumed one (trainer.py:108-114
        # builds it with last_epoch = epoch on the loaded optimizer) REQUIRES it -- the group's 'lr' is already decayed
        if last_epoch == -1:
            for g in optimizer.param_groups:
                g.setdefault('initial_lr', g['lr'])
        else:
            for i, g in enumerate(optimizer.param_groups):
                if 'initial_lr' not in g:
                    raise KeyError("param 'initial_lr' is not specified in param_groups[{}] when resuming an "
                                   "optimizer".format(i))
        self.base_lrs = [g['initial_lr'] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()

    def get_lr(self):
        if self.last_epoch > self.max_decay_steps:
            return [self.end_learning_rate for _ in self.base_lrs]
        f = (1 - self.last_epoch / self.max_decay_steps) ** self.power
        return [(b - self.end_learning_rate) * f + self.end_learning_rate for b in self.base_lrs]

    def step(self):
        self.last_epoch += 1
        for g, lr in zip(self.optimizer.param_groups, self.get_lr()):
            g['lr'] = lr

    def state_dict(self):
        return {'last_epoch': self.last_epoch, 'base_lrs': self.base_lrs}

    def load_state_dict(self, sd):
        self.last_epoch, self.base_lrs = sd['last_epoch'], sd['base_lrs']


class DataCombiCreater(object):
    """Mirror of DataCombiCreater (models/misc.py:9-128): turns a collated batch
    {'images' [B,S+1,C*,H,W], 'untrans-images', 'imus' [B,S,T,6], 'gts' [B,S+1,15]} into the
    model inputs and targets (res_imgs / res_normals [B,S,2,3,H,W], res_imu, res_gt_f2f [B,S,6],
    res_gt_f2g [B,S,7], res_gt_global).  The pair gather + channel split is one streaming HIP
    kernel per image tensor and the per-sample ground-truth Python loop one launch
    (deeplio_amd/csrc/batchprep.hip); `untrans-images` is only processed on request
    (`with_untransformed=True`), the hot loop never reads it."""

    def __init__(self, combinations, device='cpu', with_untransformed=False, c_split=3):
        """c_split: channels of the first stream (misc.py:66-68 hard-codes 0:3 | 3: -- xyz | normals of the KITTI path;
        synthetic per-stream channel counts, BASELINE's "x5", pass their own)"""
        import torch
        self.c_split = int(c_split)
        self.combinations = combinations
        self.device = torch.device(device)
        self.seq_size = get_config_container().seq_size
        self.with_untransformed = with_untransformed
        self._comb = torch.as_tensor(np.asarray(combinations), dtype=torch.int32).contiguous().to(self.device)
        self.flag = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.res_imgs = self.res_img_org = self.res_normals = self.res_normals_org = None
        self.res_imu = self.res_gt_f2f = self.res_gt_f2g = self.res_gt_global = None

    def process(self, data):
        from . import ops
        imgs = normals = imgs_org = normals_org = []
        if 'images' in data:
            imgs, normals = self.process_images(data['images'].to(self.device, non_blocking=True))
            if self.with_untransformed and 'untrans-images' in data:
                imgs_org, normals_org = self.process_images(data['untrans-images'].to(self.device, non_blocking=True))
        imus = data['imus'].to(self.device, non_blocking=True) if 'imus' in data else []
        gt_global = data['gts'].to(self.device, non_blocking=True).float().contiguous()
        gt_f2f, gt_f2g = ops.gt_relative(gt_global, self._comb, self.flag)
        self.res_imgs, self.res_normals = imgs, normals
        self.res_img_org, self.res_normals_org = imgs_org, normals_org
        self.res_imu = imus
        self.res_gt_f2f, self.res_gt_f2g, self.res_gt_global = gt_f2f, gt_f2g, gt_global

    def process_images(self, imgs):
        from . import ops
        return ops.pair_stack(imgs.float().contiguous(), self._comb, self.c_split)      # misc.py:66-68: 0:3 | 3:

    def process_ground_turth(self, gts):
        """single-sample form of the reference API: gts [S+1, 15] -> (f2f [S,6], f2g [S,7])"""
        from . import ops
        f2f, f2g = ops.gt_relative(gts.to(self.device).float().contiguous()[None], self._comb, self.flag)
        return f2f[0], f2g[0]

    def check(self):
        """raise like misc.py:106-107 if a non-finite f2f target was produced since the last check"""
        if int(self.flag.item()):
            self.flag.zero_()
            raise ValueError("gt-f2f: non-finite relative pose")

    def __call__(self, args):
        return self.process(args)
