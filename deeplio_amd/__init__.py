"""deeplio_amd -- MI355X-native (gfx950) DeepLIO training/inference hot path.

Host-side mirror of the reference's Python surface over a C-ABI HIP library:
    deeplio_amd.nets.get_model(input_shape, cfg, device)      <- deeplio.models.nets.get_model
    deeplio_amd.losses.get_loss_function(cfg, device)         <- deeplio.losses.get_loss_function
    deeplio_amd.optimizer.create_optimizer(params, cfg, args) <- deeplio.models.optimizer
    deeplio_amd.se3.se3_to_SE3(f2f_x, f2f_r)                  <- Trainer.se3_to_SE3
    deeplio_amd.misc.build_config_container(cfg, args)        <- deeplio.models.misc
Importing any of those modules loads libdeeplio_hip.so (deeplio_amd._lib) and fails loudly if
the library or one of its symbols is missing -- there is no CPU fallback.  Only
`deeplio_amd.build` and `deeplio_amd.config` are importable without the library.
"""

__version__ = "0.1.0"


_OVERLAY = ("deeplio.models.nets", "deeplio.models.misc", "deeplio.models.optimizer", "deeplio.losses")
# names the reference's worker modules bind at import time (trainer.py:17-23, tester.py:19-24,
# worker.py:13): rebound when those modules were imported before the install
_REBIND = {"nets": ("nets", None), "get_loss_function": ("losses", "get_loss_function"),
           "HWSLoss": ("losses", "HWSLoss"), "LWSLoss": ("losses", "LWSLoss"),
           "create_optimizer": ("optimizer", "create_optimizer"),
           "DataCombiCreater": ("misc", "DataCombiCreater"), "PolynomialLRDecay": ("misc", "PolynomialLRDecay"),
           "build_config_container": ("misc", "build_config_container"),
           "get_config_container": ("misc", "get_config_container")}


def install_as_deeplio(patch_workers=True):
    """Overlay this package onto the reference's module names so that the UNMODIFIED reference
    (`deeplio/train.py`, `deeplio/test.py`, `deeplio.models.trainer.Trainer`, ...) runs the HIP path:

      deeplio.models.nets / .misc / .optimizer and deeplio.losses   -> the modules of this package
      Trainer.se3_to_SE3 (trainer.py:324-351), Tester.se3_to_SE3 (tester.py:223-251)
                                                                    -> deeplio_amd.se3.se3_to_SE3

    Everything else of the reference (`deeplio.models.trainer/tester/worker`, `deeplio.datasets`,
    `deeplio.common`) stays the reference's own code and stays importable: the real `deeplio`
    package (it must be on sys.path, as it is for the reference's own entry points, train.py:8-11) is
    imported first and only the four hot-path modules are replaced inside it.  Without a reference
    checkout the four names are registered under empty stand-in packages (enough for
    `from deeplio.models import nets`).  Call it before the reference's worker modules are imported;
    if they already are, their module-level bindings are rebound as well.
    Returns the list of module names that were overlaid / patched."""
    import importlib
    import sys
    import types
    from . import losses, misc, nets, optimizer, se3
    mine = {"nets": nets, "misc": misc, "optimizer": optimizer, "losses": losses}
    try:
        importlib.import_module("deeplio")
        importlib.import_module("deeplio.models")
        have_ref = True
    except ImportError:
        have_ref = False
        for name in ("deeplio", "deeplio.models"):
            if name not in sys.modules:
                pkg = types.ModuleType(name)
                pkg.__path__ = []                     # a package, with nothing else inside
                sys.modules[name] = pkg
        sys.modules["deeplio"].models = sys.modules["deeplio.models"]
    done = []
    for name in _OVERLAY:
        parent, _, attr = name.rpartition(".")
        sys.modules[name] = mine[attr]
        setattr(sys.modules[parent], attr, mine[attr])
        done.append(name)
    if not (have_ref and patch_workers):
        return done
    for name in ("deeplio.models.worker", "deeplio.models.trainer", "deeplio.models.tester"):
        try:
            mod = importlib.import_module(name)       # resolves `nets`, `.misc`, `.optimizer` to the overlay
        except ImportError:                           # a third-party import of the worker layer is missing
            continue
        for attr, (src, member) in _REBIND.items():
            if hasattr(mod, attr):
                setattr(mod, attr, mine[src] if member is None else getattr(mine[src], member))
        if hasattr(mod, "Trainer"):
            mod.Trainer.se3_to_SE3 = lambda self, f2f_x, f2f_r: se3.se3_to_SE3(f2f_x, f2f_r)
        if hasattr(mod, "Tester"):
            mod.Tester.se3_to_SE3 = lambda self, f2f_x, f2f_r: se3.se3_to_SE3(f2f_x, f2f_r, ordering="xyzw")
        done.append(name)
    return done
