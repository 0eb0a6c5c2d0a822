"""Mirror of deeplio/losses: get_loss_function (losses/__init__.py:4-30), HWSLoss
(losses/losses.py:51-86), LWSLoss (:11-39); values and gradients come from one fused HIP
launch each (deeplio_amd.functional.PoseLossFn)."""
import torch
import torch.nn as nn

from . import functional as Fh


class _PoseLoss(nn.Module):
    mode = 0
    # 'mse' = the reference's rotation terms (losses.py:71-85); 'geodesic' = squared rotation angle
    # (BASELINE configs[4]; selected with cfg['losses']['rotation'] = 'geodesic')
    rotation = 'mse'

    def forward(self, pred_f2f_x, pred_f2f_r, pred_f2g_x, pred_f2g_r, gt_f2f_x, gt_f2f_r, gt_f2g_x,
                gt_f2g_q):
        sx = getattr(self, "sx", None)
        sq = getattr(self, "sq", None)
        mode = self.mode | (2 if self.rotation == 'geodesic' else 0)
        return Fh.PoseLossFn.apply(sx, sq, float(getattr(self, "beta", 0.)), mode,
                                   bool(self.loss_Types[0]), bool(self.loss_Types[1]), pred_f2f_x,
                                   pred_f2f_r, pred_f2g_x, pred_f2g_r, gt_f2f_x, gt_f2f_r, gt_f2g_x,
                                   gt_f2g_q)

    def __repr__(self):
        a, b = self.loss_Types
        if a and b:
            return "HWSLoss with f2f and f2g loss."
        if a:
            return "HWSLoss with only f2f loss."
        if b:
            return "HWSLoss with only f2g loss."
        return "Wrong loss combination!"


class HWSLoss(_PoseLoss):
    """Homoscedastic weighted sum: (L_p+L_t) e^-sx + sx + (L_q+L_w) e^-sq + sq"""
    mode = 0

    def __init__(self, sx=0., sq=-2.5, learn_hyper_params=True, device="cpu", loss_Types=(True, True)):
        super().__init__()
        self.learn_hyper_params = learn_hyper_params
        self.loss_Types = list(loss_Types)
        self.sx = nn.Parameter(torch.tensor(float(sx), device=device), requires_grad=learn_hyper_params)
        self.sq = nn.Parameter(torch.tensor(float(sq), device=device), requires_grad=learn_hyper_params)


class LWSLoss(_PoseLoss):
    """Linear weighted sum: (L_p+L_t) + beta (L_q+L_w)"""
    mode = 1

    def __init__(self, beta=1125., gamma=1., loss_Types=(True, True)):
        super().__init__()
        self.beta, self.gamma = beta, gamma
        self.loss_Types = list(loss_Types)


def get_loss_function(cfg, device):
    loss_cfg = cfg['losses']
    loss_name = loss_cfg['active'].lower()
    params = loss_cfg.get(loss_name, {}).get('params', {})
    loss_type = loss_cfg['loss-type'].lower()
    if "+" in loss_type:
        loss_types = [True, True]
    elif loss_type == "global":
        loss_types = [False, True]
    elif loss_type == "local":
        loss_types = [True, False]
    else:
        raise ValueError("Wrong loss type selected!")
    rotation = str(loss_cfg.get('rotation', 'mse')).lower()
    if rotation not in ('mse', 'geodesic'):
        raise ValueError("Rotation loss {} is not supported!".format(rotation))
    if loss_name == 'hwsloss':
        loss = HWSLoss(sx=params.get('sx', 0.), sq=params.get('sq', -2.5),
                       learn_hyper_params=params.get('learn', False), device=device,
                       loss_Types=loss_types)
    elif loss_name == 'lwsloss':
        loss = LWSLoss(beta=params.get('beta', 1125.), loss_Types=loss_types)
    else:
        raise ValueError("Loss {} is not supported!".format(loss_name))
    loss.rotation = rotation
    return loss
