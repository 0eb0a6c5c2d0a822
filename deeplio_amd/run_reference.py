"""Run one of the reference's own entry points on the HIP path without editing it:

    python -m deeplio_amd.run_reference /path/to/DeepLIO/deeplio/train.py -b 8 --device cuda -c config.yaml
    python -m deeplio_amd.run_reference /path/to/DeepLIO/deeplio/test.py  --device cuda -c config.yaml

Puts the reference checkout on sys.path the way its scripts do (train.py:8-11), overlays
deeplio.models.nets / .misc / .optimizer / deeplio.losses and the two se3_to_SE3 methods with this
package (deeplio_amd.install_as_deeplio) and then executes the script as __main__ with the
remaining arguments (the flags of train.py:28-57 / test.py:28-42 are the reference's own)."""
import os
import runpy
import sys


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv or argv[0] in ("-h", "--help"):
        print(__doc__)
        return 2
    script = os.path.abspath(argv[0])
    if not os.path.isfile(script):
        raise SystemExit("run_reference: %s is not a file" % script)
    pkg_dir = os.path.dirname(script)                       # .../deeplio
    for p in (pkg_dir, os.path.dirname(pkg_dir)):
        if p not in sys.path:
            sys.path.append(p)
    import deeplio_amd
    deeplio_amd.install_as_deeplio()
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name="__main__")
    return 0


if __name__ == "__main__":
    sys.exit(main())
