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


def test_cooperative_batchnorm_mode_switch_needs_no_device():
    """dlio_bn_coop_set_mode / _get_mode (host-side state only): the default is mode 3 -- one item per workgroup where
    concurrent launches cannot fill an XCD with waiting workgroups, persistent otherwise -- unless the environment
    says otherwise; out-of-range values are refused"""
    from deeplio_amd import _lib
    lib = _lib.lib
    default = int(os.environ.get("DLIO_BN_COOP_MODE", "3"))
    assert lib.dlio_bn_coop_get_mode() == default
    try:
        for m in (0, 1, 2, 3):
            assert lib.dlio_bn_coop_set_mode(m) == 0 and lib.dlio_bn_coop_get_mode() == m
        assert lib.dlio_bn_coop_set_mode(4) != 0 and lib.dlio_bn_coop_set_mode(-2) != 0
        assert lib.dlio_bn_coop_get_mode() == 3
    finally:
        assert lib.dlio_bn_coop_set_mode(-1) == 0
    assert lib.dlio_bn_coop_get_mode() == default


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


def test_polynomial_lr_decay_resumes_like_a_torch_scheduler():
    """trainer.py:108-114 rebuilds the schedule with last_epoch = epoch on the LOADED optimizer: the resumed schedule
    continues the uninterrupted one (base rate = 'initial_lr', not the already-decayed 'lr'), the optimizer state written
    through the overlay carries 'initial_lr' (torch's own _LRScheduler raises KeyError without it), and a checkpoint
    torch.optim + a torch scheduler wrote resumes here"""
    from deeplio_amd.misc import PolynomialLRDecay

    def fresh():
        return torch.optim.Adam([torch.nn.Parameter(torch.zeros(3))], lr=1e-3)

    opt = fresh()
    sch = PolynomialLRDecay(opt, max_decay_steps=30, end_learning_rate=1e-6, power=2.0)
    table = []
    for _ in range(30):
        table.append(opt.param_groups[0]['lr'])
        sch.step()
    # interrupted at epoch 15: save, load into a new optimizer, rebuild the schedule as the trainer does
    opt = fresh()
    sch = PolynomialLRDecay(opt, max_decay_steps=30, end_learning_rate=1e-6, power=2.0)
    for _ in range(15):
        sch.step()
    sd = opt.state_dict()
    assert sd['param_groups'][0]['initial_lr'] == 1e-3
    opt2 = fresh()
    opt2.load_state_dict(sd)
    sch2 = PolynomialLRDecay(opt2, max_decay_steps=30, end_learning_rate=1e-6, power=2.0, last_epoch=15)
    resumed = []
    for _ in range(14):
        resumed.append(opt2.param_groups[0]['lr'])
        sch2.step()
    assert np.allclose(resumed, table[16:30], rtol=1e-12)
    # the same state resumes under a stock torch scheduler (it needs 'initial_lr' in the groups) ...
    opt3 = fresh()
    opt3.load_state_dict(sd)
    torch.optim.lr_scheduler.StepLR(opt3, 10, last_epoch=15)
    # ... and a resume without it fails the way torch's does
    bare = types.SimpleNamespace(param_groups=[{'lr': 1e-3}])
    with pytest.raises(KeyError):
        PolynomialLRDecay(bare, max_decay_steps=30, last_epoch=3)


def _run_py(code, cwd):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([ROOT, os.path.join(HERE, "golden")]))
    return subprocess.run([sys.executable, "-c", code], cwd=str(cwd), env=env, capture_output=True, text=True,
                          timeout=600)


def test_install_as_deeplio_without_a_reference_checkout(tmp_path):
    """no `deeplio` on sys.path: the four hot-path names resolve to this package"""
    code = r"""
import sys
import deeplio_amd
done = deeplio_amd.install_as_deeplio()
from deeplio.models import nets as rn
from deeplio import losses as rl
from deeplio.models.misc import build_config_container, PolynomialLRDecay, DataCombiCreater
from deeplio.models.optimizer import create_optimizer
assert rn.get_model is deeplio_amd.nets.get_model and rl.get_loss_function is deeplio_amd.losses.get_loss_function
assert create_optimizer is deeplio_amd.optimizer.create_optimizer
print("OK", sorted(done))
"""
    r = _run_py(code, tmp_path)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


FAKE_TREE = {       # a stand-in checkout with the reference's import STRUCTURE (train.py:8-13, trainer.py:17-23,
    # worker.py:12-13, tester.py) -- written for this test, nothing of the reference's code
    "deeplio/__init__.py": "",
    "deeplio/common/__init__.py": "LOGGER = 'reference logger'\n",
    "deeplio/datasets/__init__.py": "class Kitti:\n    origin = 'reference dataset'\n",
    "deeplio/losses/__init__.py": "def get_loss_function(cfg, device):\n    raise RuntimeError('reference loss')\n"
                                  "HWSLoss = LWSLoss = object\n",
    "deeplio/models/__init__.py": "",
    "deeplio/models/nets/__init__.py": "def get_model(input_shape, cfg, device):\n    raise RuntimeError('reference nets')\n",
    "deeplio/models/misc.py": "def build_config_container(cfg, args):\n    raise RuntimeError('reference misc')\n"
                              "DataCombiCreater = PolynomialLRDecay = object\n",
    "deeplio/models/optimizer.py": "def create_optimizer(params, cfg, args):\n    raise RuntimeError('reference optimizer')\n",
    "deeplio/models/worker.py": "from deeplio.common import *\nfrom .misc import build_config_container\n"
                                "class Worker:\n    def __init__(self, args, cfg):\n"
                                "        self.cc = build_config_container(cfg, args)\n",
    "deeplio/models/trainer.py": "from deeplio import datasets as ds\nfrom deeplio.common import LOGGER\n"
                                 "from deeplio.losses import get_loss_function, HWSLoss, LWSLoss\n"
                                 "from deeplio.models import nets\n"
                                 "from deeplio.models.misc import DataCombiCreater, PolynomialLRDecay\n"
                                 "from .optimizer import create_optimizer\nfrom .worker import Worker\n"
                                 "class Trainer(Worker):\n"
                                 "    def __init__(self, args, cfg, shape):\n"
                                 "        super().__init__(args, cfg)\n"
                                 "        self.model = nets.get_model(input_shape=shape, cfg=cfg, device=self.cc.device)\n"
                                 "        self.criterion = get_loss_function(cfg, args.device)\n"
                                 "        self.dataset = ds.Kitti()\n"
                                 "    def se3_to_SE3(self, x, r):\n        raise RuntimeError('reference se3')\n"
                                 "class TrainerDeepLIO(Trainer):\n    pass\n",
    "deeplio/train.py": "import os, sys\n"
                        "dname = os.path.abspath(os.path.dirname(__file__))\n"
                        "sys.path.append(dname)\nsys.path.append(os.path.abspath(dname + '/..'))\n"
                        "from deeplio.models.trainer import TrainerDeepLIO\n"
                        "if __name__ == '__main__':\n"
                        "    import types, json\n"
                        "    sys.path.insert(0, os.environ['GOLDEN_DIR'])\n"
                        "    import golden_common as gc\n"
                        "    cfg = gc.case_cfg('pointseg_lstm_cat')\n"
                        "    args = types.SimpleNamespace(device='cpu', batch_size=2, flags=sys.argv[1:])\n"
                        "    t = TrainerDeepLIO(args, cfg, (5, 16, 64))\n"
                        "    print('TRAINER', type(t.model).__module__, type(t.criterion).__module__, t.dataset.origin,\n"
                        "          'fc_pos.weight' in t.model.state_dict(), args.flags)\n",
}


def _write_tree(root, tree):
    for rel, text in tree.items():
        f = root / rel
        f.parent.mkdir(parents=True, exist_ok=True)
        f.write_text(text)


def test_run_reference_overlays_an_unmodified_entry_point(tmp_path):
    """python -m deeplio_amd.run_reference <checkout>/deeplio/train.py <flags>: the script's own
    `from deeplio.models.trainer import TrainerDeepLIO` (train.py:13) resolves to the CHECKOUT's
    trainer, whose nets / losses / misc / optimizer are this package; datasets / common stay the
    checkout's.  Uses a stand-in checkout with the reference's import structure (the reference's
    real worker layer needs KITTI on disk)."""
    _write_tree(tmp_path, FAKE_TREE)
    env = dict(os.environ, PYTHONPATH=ROOT, GOLDEN_DIR=os.path.join(HERE, "golden"))
    r = subprocess.run([sys.executable, "-m", "deeplio_amd.run_reference", str(tmp_path / "deeplio" / "train.py"),
                        "-b", "8", "--device", "cuda"], cwd=str(tmp_path), env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [l for l in r.stdout.splitlines() if l.startswith("TRAINER")][0]
    assert "deeplio_amd.nets deeplio_amd.losses reference dataset True" in line and "'-b', '8'" in line, line
    # the order train.py itself would produce when a maintainer adds the two lines at its top:
    # install AFTER the worker layer was imported rebinds the module-level names as well
    code = r"""
import sys
sys.path.insert(0, %r)
import deeplio.models.trainer as tr
import deeplio_amd
done = deeplio_amd.install_as_deeplio()
assert tr.nets is deeplio_amd.nets and tr.create_optimizer is deeplio_amd.optimizer.create_optimizer
assert tr.get_loss_function is deeplio_amd.losses.get_loss_function
import deeplio.models.worker as wk
assert wk.build_config_container is deeplio_amd.misc.build_config_container
import deeplio.datasets as ds
assert ds.Kitti.origin == 'reference dataset'
print("OK", done)
""" % str(tmp_path)
    r = _run_py(code, tmp_path)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/deeplio"), reason="needs the reference checkout (build container)")
def test_install_as_deeplio_overlays_the_real_reference(tmp_path):
    """With the REAL reference on sys.path (third-party imports it lacks here stubbed as in
    make_golden.install_stubs): after the install the reference's own `deeplio.models.trainer`
    imports (train.py:13), its `nets` / `get_loss_function` / `create_optimizer` are this package, the
    model is constructed through the very call of trainer.py:52 (`nets.get_model(input_shape=..., cfg=...,
    device=...)` after worker.py:41's `build_config_container`), `Trainer.se3_to_SE3` is the HIP chain,
    and `deeplio.datasets` / `deeplio.common` / `deeplio.models.tester` are still the reference's."""
    code = r"""
import os, sys, types
import make_golden as mg
mg.install_stubs()
sys.path.insert(0, mg.REF)
import deeplio_amd
done = deeplio_amd.install_as_deeplio()
from deeplio.models.trainer import TrainerDeepLIO            # train.py:13, unmodified
import deeplio.models.trainer as tr, deeplio.models.worker as wk, deeplio.models.tester as te
import deeplio.datasets as ds, deeplio.common.spatial as sp
for m in (tr, wk, te, ds, sp):
    assert m.__file__.startswith(mg.REF), m.__file__
assert tr.nets is deeplio_amd.nets and tr.get_loss_function is deeplio_amd.losses.get_loss_function
assert tr.create_optimizer is deeplio_amd.optimizer.create_optimizer
assert tr.PolynomialLRDecay is deeplio_amd.misc.PolynomialLRDecay and tr.DataCombiCreater is deeplio_amd.misc.DataCombiCreater
assert wk.build_config_container is deeplio_amd.misc.build_config_container
import golden_common as gc
cfg = gc.case_cfg('pointseg_lstm_cat')
args = types.SimpleNamespace(device='cpu', batch_size=2)
cc = wk.build_config_container(cfg, args)                     # worker.py:41
model = tr.nets.get_model(input_shape=(5, 16, 64), cfg=cfg, device=cc.device)      # trainer.py:52
crit = tr.get_loss_function(cfg, args.device)                 # trainer.py:55
assert type(model).__module__ == 'deeplio_amd.nets' and type(crit).__module__ == 'deeplio_amd.losses'
assert model.name == 'deeplio' and 'lidar_feat_net.encoder1.fire_blk1.0.squeeze_bn.running_mean' in model.state_dict()
import torch
try:
    tr.Trainer.se3_to_SE3(types.SimpleNamespace(device='cpu'), torch.zeros(1, 2, 3), torch.zeros(1, 2, 3))
    raise SystemExit("the patched se3_to_SE3 should refuse CPU tensors")
except RuntimeError as e:
    assert "no CPU fallback" in str(e), e
print("OK", done)
"""
    r = _run_py(code, tmp_path)
    assert r.returncode == 0 and "OK" in r.stdout, r.stdout + r.stderr
    assert "deeplio.models.trainer" in r.stdout and "deeplio.models.tester" in r.stdout


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


def test_stale_library_is_rejected_at_import(tmp_path):
    """_lib._load() compares the library's ABI version / header CRC with include/deeplio_hip.h: a
    library built from another revision of the header must fail at import, not at a call"""
    from deeplio_amd import _header, _lib
    assert _lib.lib.dlio_version() == _header.abi_version() and _lib.lib.dlio_abi_hash() == _header.abi_hash()
    hdr = tmp_path / "deeplio_hip.h"
    hdr.write_text(open(_header.HEADER).read().replace("int dlio_version(void);", "long dlio_version(void);"))
    assert _header.abi_hash(str(hdr)) != _header.abi_hash()
    code = ("import deeplio_amd._header as h; h.abi_hash = lambda path=None: 12345\n"
            "try:\n    import deeplio_amd._lib\nexcept ImportError as e:\n    print('REJECTED', e)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert "REJECTED" in r.stdout and "different include/deeplio_hip.h" in r.stdout, r.stdout + r.stderr


def test_library_built_with_a_timing_probe_is_rejected_at_import():
    """the product library carries no timing probe (-DDLIO_SPLIT_Q0 / -DBX3_ABLATE / -DW1_COAL_PROBE build kernels that skip
    work and give wrong results: tools/variant_lib.py); a library that reports one is refused at import unless
    DLIO_ALLOW_PROBES=1"""
    from deeplio_amd import _lib
    assert _lib.lib.dlio_build_probes() == 0
    code = ("import ctypes, deeplio_amd._lib as L\n")
    # simulate a probe build: patch the table's loader so that the probe query answers 1
    code = ("import ctypes as C\n"
            "real = C.CDLL\n"
            "class Fake:\n"
            "    def __init__(self, path): self._l = real(path)\n"
            "    def __getattr__(self, n):\n"
            "        if n == 'dlio_build_probes':\n"
            "            f = lambda: 1\n"
            "            return type('F', (), {'__call__': staticmethod(f), 'restype': None, 'argtypes': None})()\n"
            "        return getattr(self._l, n)\n"
            "C.CDLL = Fake\n"
            "try:\n    import deeplio_amd._lib\nexcept ImportError as e:\n    print('REJECTED', e)\n")
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True)
    assert "REJECTED" in r.stdout and "timing-probe" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, "-c", code], cwd=ROOT, capture_output=True, text=True,
                       env=dict(os.environ, DLIO_ALLOW_PROBES="1"))
    assert "REJECTED" not in r.stdout, r.stdout + r.stderr


def test_oracle_geodesic_rotation_terms_known_answers():
    """the geodesic rotation terms (BASELINE configs[4]; definition in include/deeplio_hip.h): angle
    between rotations about one axis = difference of the angles; invariant to quaternion sign, scale
    and component order; equals |log(Ra^T Rb)| of the SO(3) restatement"""
    from oracle import model as om
    from oracle import se3
    z = torch.tensor([0., 0., 1.], dtype=torch.float64)
    for a, b in ((0.3, 0.0), (0.3, -0.5), (3.0, 0.1), (1e-6, 0.)):
        qa, qb = om.so3_to_quat(a * z), om.so3_to_quat(b * z)
        assert abs(float(om.geodesic_theta2(qa, qb)) - (a - b) ** 2) < 1e-12
        assert abs(float(om.geodesic_theta2(-3. * qa, qb)) - (a - b) ** 2) < 1e-12
        assert abs(float(om.geodesic_theta2(qa[[1, 2, 3, 0]], qb[[1, 2, 3, 0]])) - (a - b) ** 2) < 1e-12
    g = torch.Generator().manual_seed(0)
    wa, wb = torch.randn(16, 3, generator=g, dtype=torch.float64), torch.randn(16, 3, generator=g, dtype=torch.float64)
    th2 = om.geodesic_theta2(om.so3_to_quat(wa), om.so3_to_quat(wb))
    for i in range(16):
        rel = se3.so3_exp(wa[i]).t() @ se3.so3_exp(wb[i])
        assert abs(float(th2[i]) - float(se3.so3_log(rel).norm() ** 2)) < 1e-9
        assert abs(float(se3.rot_to_quat(se3.so3_exp(wa[i])) @ om.so3_to_quat(wa[i])).__abs__() - 1.) < 1e-12
    # identical rotations: zero loss and a finite (zero) gradient
    q = om.so3_to_quat(wa).requires_grad_(True)
    v = om.geodesic_theta2(q, q.detach()).sum()
    v.backward()
    assert float(v) < 1e-20 and torch.isfinite(q.grad).all()


def test_bench_gpus_n_starts_its_own_ranks():
    """`python bench.py --gpus 2` without torchrun spawns two ranks (gloo here: --dry-run skips the
    model, which needs the GPU), rank 0 prints ONE JSON line with n_gpus = 2; a WORLD_SIZE that
    contradicts --gpus and a missing GPU both fail loudly"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=env, capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout + r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["dry_run"] is True and d["rank_sum"] == 3.0
    r = subprocess.run([sys.executable, bench, "--gpus", "2", "--dry-run"], env=dict(env, WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
    if not torch.cuda.is_available():
        r = subprocess.run([sys.executable, bench, "--gpus", "2", "--steps", "1"], env=env, capture_output=True,
                           text=True, timeout=600, cwd=ROOT)
        assert r.returncode != 0 and "HIP device(s) visible" in r.stderr
