"""Functional CPU references for the per-op parity tests: the reference's arithmetic IS
stock torch.nn / torch.nn.functional on CPU, so these are thin, explicit
restatements of the calls made at the cited reference sites, evaluated in fp64 where the
test wants a tighter yardstick.  TEST INFRASTRUCTURE (see oracle/__init__.py)."""
import torch
import torch.nn.functional as F


def conv2d(x, w, b, stride, padding):
    """nn.Conv2d forward (pointseg_modules.py:96-106, base_net.py:59, ...)."""
    return F.conv2d(x, w, b, stride=stride, padding=padding)


def bn_train(x, gamma, beta, rm, rv, momentum, eps):
    """nn.BatchNorm2d in training mode; returns y and updated running stats."""
    rm, rv = rm.clone(), rv.clone()
    y = F.batch_norm(x, rm, rv, gamma, beta, True, momentum, eps)
    return y, rm, rv


def bn_eval(x, gamma, beta, rm, rv, eps):
    return F.batch_norm(x, rm, rv, gamma, beta, False, 0.0, eps)


def maxpool(x, k, stride, padding, ceil_mode=False):
    return F.max_pool2d(x, k, stride, padding, ceil_mode=ceil_mode)


def gap(x):
    return F.adaptive_avg_pool2d(x, (1, 1)).flatten(1)


ACTS = {0: lambda v: v, 1: F.relu, 2: lambda v: F.leaky_relu(v, 0.01), 3: torch.sigmoid,
        4: torch.tanh}


def linear(x, w, b, act=0):
    return ACTS[act](F.linear(x, w, b))
