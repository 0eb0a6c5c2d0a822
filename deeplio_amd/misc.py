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
        self.base_lrs = [g['lr'] for g in optimizer.param_groups]
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
