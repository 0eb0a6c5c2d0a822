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


def install_as_deeplio():
    """Register this package under the reference's module names so that unmodified
    `from deeplio.models import nets` / `from deeplio import losses` resolve here."""
    import sys
    import types
    from . import losses, misc, nets, optimizer
    root = types.ModuleType("deeplio")
    models = types.ModuleType("deeplio.models")
    models.nets, models.misc, models.optimizer = nets, misc, optimizer
    root.models, root.losses = models, losses
    sys.modules.update({"deeplio": root, "deeplio.models": models, "deeplio.models.nets": nets,
                        "deeplio.models.misc": misc, "deeplio.models.optimizer": optimizer,
                        "deeplio.losses": losses})
