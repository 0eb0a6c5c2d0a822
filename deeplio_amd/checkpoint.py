"""Checkpoint layout of the reference: `trainer.py:156-170, 439-444` writes

    <dir>/cpkt_<model.name>.tar        {'epoch','state_dict','best_acc','optimizer','criterion'}
    <dir>/cpkt_<feat_net.name>.tar     {'state_dict'}                 (one per sub-network)
    <dir>/..._best.tar                 copies when the validation loss improved

with `torch.save`; `nets/__init__.py:216-223` loads `['state_dict']`.  Module state_dict keys and
shapes are the reference's already (deeplio_amd/nets.py), so model/criterion entries are
interchangeable as they are.  The optimizer entry is converted between the flat-buffer optimizer
and `torch.optim.{Adam,SGD,RMSprop,Adadelta}.state_dict()` ({'state': {i: {'step','exp_avg','exp_avg_sq'}},
'param_groups': [...]}), so a reference run can be resumed here and vice versa."""
import os
import shutil

import torch


def save_checkpoint(state, is_best, checkpoint_dir, filename="checkpoint"):
    """trainer.py:439-444"""
    file_path = '{}/{}.tar'.format(checkpoint_dir, filename)
    torch.save(state, file_path)
    if is_best:
        shutil.copyfile(file_path, '{}/{}_best.tar'.format(checkpoint_dir, filename))
    return file_path


def _torch_hyper(opt):
    """the per-group hyper-parameters torch.optim would list for this optimizer class"""
    kind = type(opt).__name__
    if kind == 'Adam':
        return dict(betas=tuple(opt.betas), eps=opt.eps, amsgrad=False)
    if kind == 'SGD':
        return dict(momentum=opt.momentum, dampening=0, nesterov=False)
    if kind == 'RMSprop':
        return dict(alpha=opt.alpha, eps=opt.eps, momentum=opt.momentum, centered=opt.centered)
    if kind == 'Adadelta':
        return dict(rho=opt.rho, eps=opt.eps)
    return {}


def optimizer_to_torch_state(opt):
    """FlatOptimizer -> the dict torch.optim.{Adam,SGD,RMSprop,Adadelta}.state_dict() would hold
    (the flat state buffers carry torch's own state key names: exp_avg, exp_avg_sq,
    momentum_buffer, square_avg, grad_avg, acc_delta)"""
    state, groups, idx = {}, [], 0
    index_of = {id(p): i for i, p in enumerate(opt.params)}
    bufs = opt._state()
    with_step = type(opt).__name__ != 'SGD'
    for g in opt.param_groups:
        ids = []
        for p in g['params']:
            i = index_of[id(p)]
            o = opt.offsets[i]
            if opt.step_count > 0:
                st = {k: b[o:o + p.numel()].view(p.shape).detach().clone() for k, b in bufs.items()}
                if with_step:
                    st['step'] = torch.tensor(float(opt.step_count))
                state[idx] = st
            ids.append(idx)
            idx += 1
        hyper = {k: v for k, v in g.items() if k != 'params'}
        hyper.update(_torch_hyper(opt))
        hyper['params'] = ids
        groups.append(hyper)
    return {'state': state, 'param_groups': groups}


def optimizer_from_torch_state(opt, sd):
    """load a torch.optim state_dict (or one written by optimizer_to_torch_state)"""
    index_of = {id(p): i for i, p in enumerate(opt.params)}
    bufs = opt._state()
    step = 0
    with torch.no_grad():
        for g, sg in zip(opt.param_groups, sd['param_groups']):
            # every hyper-parameter entry of the group ('lr', 'weight_decay', and 'initial_lr', which the LR
            # schedule needs to resume from the un-decayed rate) -- 'params' holds indices, not values
            for k, v in sg.items():
                if k != 'params':
                    g[k] = v
            for p, sid in zip(g['params'], sg['params']):
                st = sd['state'].get(sid)
                if st is None:
                    continue
                i = index_of[id(p)]
                o = opt.offsets[i]
                for k, b in bufs.items():
                    if st.get(k) is not None:
                        b[o:o + p.numel()].view(p.shape).copy_(st[k])
                        step = max(step, 1)
                if 'step' in st:
                    step = max(step, int(st['step']))
    if 'step_count' in sd:
        step = sd['step_count']
    opt.step_count = int(step)


def save_training_state(checkpoint_dir, epoch, model, criterion, optimizer, best_acc, is_best):
    """trainer.py:156-170: the model file plus one file per sub-network"""
    os.makedirs(checkpoint_dir, exist_ok=True)
    files = [save_checkpoint({'epoch': epoch, 'state_dict': model.state_dict(), 'best_acc': best_acc,
                              'optimizer': optimizer_to_torch_state(optimizer),
                              'criterion': criterion.state_dict()},
                             is_best, checkpoint_dir, 'cpkt_{}'.format(model.name))]
    for feat_net in model.get_feat_networks():
        files.append(save_checkpoint({'state_dict': feat_net.state_dict()}, is_best, checkpoint_dir,
                                     'cpkt_{}'.format(feat_net.name)))
    return files


def load_training_state(path, model, criterion=None, optimizer=None, map_location=None):
    """-> (epoch, best_acc).  Accepts files written by the reference or by save_training_state."""
    if not os.path.isfile(path):
        raise FileNotFoundError("No checkpoint found ({})!".format(path))
    sd = torch.load(path, map_location=map_location or model.device, weights_only=False)
    model.load_state_dict(sd['state_dict'])
    if criterion is not None and 'criterion' in sd:
        criterion.load_state_dict(sd['criterion'])
    if optimizer is not None and 'optimizer' in sd:
        optimizer_from_torch_state(optimizer, sd['optimizer'])
    return sd.get('epoch', 0), sd.get('best_acc', float('inf'))
