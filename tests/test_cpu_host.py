"""CPU tests (-m "not gpu"): the C-ABI library loads and exports every symbol the header
declares, the ctypes table matches the prototypes, host logic mirrors the reference's
surface (factory errors, config container, LR schedule, state_dict keys), the product path
refuses to run without the GPU, and the data-parallel exchange works over gloo (world 2)."""
import ast
import os
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(HERE, "golden"))
import golden_common as gc  # noqa: E402


def test_library_exports_every_declared_symbol():
    from deeplio_amd import _lib
    from deeplio_amd._header import prototypes
    protos = prototypes()
    assert len(protos) >= 50
    for name, nargs in protos.items():
        assert hasattr(_lib.lib, name), name                    # exported by the .so
        assert name in _lib.SIGNATURES, name                    # bound by the host
        assert len(_lib.SIGNATURES[name][1]) == nargs, name     # same arity as the header
    assert set(_lib.SIGNATURES) == set(protos)
    assert _lib.lib.dlio_arch() == b"gfx950" and _lib.lib.dlio_version() >= 100
    assert _lib.strerror(-2) == "unsupported configuration"


def test_error_codes_map_to_reference_style_exceptions():
    from deeplio_amd import _lib
    with pytest.raises(ValueError):
        _lib.check(_lib.DLIO_EINVAL, "x")
    with pytest.raises(ValueError):
        _lib.check(_lib.DLIO_EUNSUP, "x")
    with pytest.raises(RuntimeError):
        _lib.check(_lib.DLIO_EWS, "x")
    # argument validation happens before any launch: no GPU needed
    assert _lib.lib.dlio_conv2d_fwd(None, None, None, None, None, None, None, None, None, None) == _lib.DLIO_EINVAL
    assert _lib.lib.dlio_adam_step(None, None, None, None, 0, 0., 0., 0., 0., 0., 1, 1., None) == _lib.DLIO_EINVAL


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deeplio_amd")
    for fn in os.listdir(pkg):
        if not fn.endswith(".py"):
            continue
        tree = ast.parse(open(os.path.join(pkg, fn)).read())
        for node in ast.walk(tree):
            names = []
            if isinstance(node, ast.Import):
                names = [a.name for a in node.names]
            elif isinstance(node, ast.ImportFrom):
                names = [node.module or ""]
            assert not any(n.split(".")[0] == "oracle" for n in names), (fn, names)


def test_product_path_fails_loudly_without_gpu():
    from deeplio_amd import misc, nets, ops
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    cfg = gc.case_cfg("pointseg_lstm_cat")
    misc.build_config_container(cfg, types.SimpleNamespace(device="cpu", batch_size=2))
    model = nets.get_model((5, 16, 64), cfg, "cpu")            # construction is host-only
    xyz = torch.randn(1, 2, 2, 5, 16, 64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model([[xyz, xyz], torch.rand(1, 2, 5, 6)])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.linear_fwd(torch.randn(2, 4), torch.randn(3, 4), None)


@pytest.mark.parametrize("name", list(gc.MODEL_CASES))
def test_state_dict_keys_and_shapes_match_reference_layout(name):
    """keys/shapes equal the oracle's, which is pinned to the reference by the goldens"""
    from deeplio_amd import misc, nets
    from oracle import model as om
    g = gc.MODEL_CASES[name]['geom']
    cfg = gc.case_cfg(name)
    misc.build_config_container(cfg, types.SimpleNamespace(device="cpu", batch_size=2))
    m = nets.get_model((g['C'], g['H'], g['W']), cfg, "cpu")
    o = om.get_model((g['C'], g['H'], g['W']), cfg)
    a, b = m.state_dict(), o.state_dict()
    assert list(a.keys()) == list(b.keys())
    assert all(a[k].shape == b[k].shape for k in a)
    assert m.name == "deeplio" and [n.name for n in m.get_feat_networks()] == [n.name for n in o.get_feat_networks()]
    assert list(m.lidar_feat_net.get_output_shape()) == [1, g['S'], 128]


def test_headline_model_has_reference_parameter_count():
    from deeplio_amd import misc, nets
    from deeplio_amd.config import make_config
    cfg = make_config(seq=5)
    misc.build_config_container(cfg, types.SimpleNamespace(device="cpu", batch_size=2))
    m = nets.get_model((3, 57, 720), cfg, "cpu")                # shipped default geometry
    assert len(m.state_dict()) == 576                           # 576 entries
    assert sum(p.numel() for p in m.odom_feat_net.parameters()) == 35684352


def test_factory_errors_follow_reference():
    from deeplio_amd import losses, misc, nets
    from deeplio_amd.config import make_config
    from deeplio_amd.optimizer import create_optimizer
    with pytest.raises(ValueError, match="Config container"):
        misc.config_container = None
        nets.get_model((3, 16, 64), make_config(), "cpu")
    cfg = make_config(lidar="lidar-feat-bogus")
    cfg['lidar-feat-bogus'] = {}
    misc.build_config_container(cfg, types.SimpleNamespace(device="cpu", batch_size=2))
    with pytest.raises(ValueError, match="Wrong feature network"):
        nets.get_model((3, 16, 64), cfg, "cpu")
    cfg = make_config(odom="odom-feat-bogus")
    with pytest.raises(ValueError, match="Wrong odometry feature network"):
        nets.get_model((3, 16, 64), cfg, "cpu")
    cfg = make_config()
    cfg['losses']['loss-type'] = "nonsense"
    with pytest.raises(ValueError, match="Wrong loss type"):
        losses.get_loss_function(cfg, "cpu")
    cfg['losses']['loss-type'] = "local"
    cfg['losses']['active'] = "geoloss"
    with pytest.raises(ValueError, match="not supported"):
        losses.get_loss_function(cfg, "cpu")
    cfg = make_config()
    cfg['optimizer'] = "lion"
    with pytest.raises(ValueError, match="not supported"):
        create_optimizer([torch.nn.Parameter(torch.zeros(1))], cfg, types.SimpleNamespace(lr=1., weight_decay=0., momentum=0.))
    crit = losses.get_loss_function(make_config(), "cpu")
    assert crit.loss_Types == [True, True] and float(crit.sx) == 0. and float(crit.sq) == -3.


def test_polynomial_lr_decay_matches_reference_table():
    from deeplio_amd.misc import PolynomialLRDecay
    gold = np.load(os.path.join(HERE, "golden", "gt_lr.npz"))
    opt = types.SimpleNamespace(param_groups=[{'lr': 1e-3}])
    sch = PolynomialLRDecay(opt, max_decay_steps=30, end_learning_rate=1e-6, power=2.0)
    lrs = []
    for _ in range(33):
        lrs.append(opt.param_groups[0]['lr'])
        sch.step()
    assert np.allclose(lrs, gold['lr_table'], rtol=1e-6)


def test_install_as_deeplio_aliases():
    import deeplio_amd
    deeplio_amd.install_as_deeplio()
    from deeplio.models import nets as rn
    from deeplio import losses as rl
    assert rn.get_model is deeplio_amd.nets.get_model and rl.get_loss_function is deeplio_amd.losses.get_loss_function
    for k in [k for k in sys.modules if k == "deeplio" or k.startswith("deeplio.")]:
        del sys.modules[k]


DP_WORKER = r'''
import os, sys, torch
sys.path.insert(0, %r)
import torch.distributed as dist
from deeplio_amd import dist as ddist
world, rank, local = ddist.init("gloo")
assert world == 2
flat = torch.full((1000,), float(rank + 1))
grad = torch.arange(1000, dtype=torch.float32) * (rank + 1)
opt = type("O", (), {"grad_scale": 1.0})()
sync = ddist.GradSync(flat, grad, opt)
sync.broadcast_parameters()
assert torch.equal(flat, torch.full((1000,), 1.0))          # rank 0's parameters everywhere
sync.all_reduce_grads()
assert torch.equal(grad, torch.arange(1000, dtype=torch.float32) * 3) and opt.grad_scale == 0.5
# two buckets: the tail is reduced asynchronously (as from the backward hook), the head afterwards
grad2 = torch.arange(1000, dtype=torch.float32) * (rank + 1)
sync2 = ddist.GradSync(flat, grad2, opt)
sync2.set_tail(640)
assert sync2.tail_lo == 640
sync2.reduce_tail_async()
sync2.reduce_tail_async()                                   # idempotent within a step
sync2.all_reduce_grads()
assert torch.equal(grad2, torch.arange(1000, dtype=torch.float32) * 3) and sync2._tail_work is None
grad2.copy_(torch.arange(1000, dtype=torch.float32) * (rank + 1))
sync2.all_reduce_grads()                                    # hook did not fire: one full all-reduce
assert torch.equal(grad2, torch.arange(1000, dtype=torch.float32) * 3)
sync2.set_tail(0); assert sync2.tail_lo is None
assert list(ddist.shard_batch(8, world, rank)) == list(range(rank * 4, rank * 4 + 4))
assert sync.max_over_ranks(float(rank)) == 1.0
dist.barrier(); dist.destroy_process_group()
print("rank", rank, "ok")
'''


def test_data_parallel_exchange_gloo_world2(tmp_path):
    script = tmp_path / "dp_worker.py"
    script.write_text(DP_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert "rank 0 ok" in outs[0] and "rank 1 ok" in outs[1]
