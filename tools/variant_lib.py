#!/usr/bin/env python3
"""A/B of compile-time switches: build a variant of libdeeplio_hip.so with extra -D flags, run a command with it.

    python tools/variant_lib.py build noswz -DDLIO_XCD_SWIZZLE=0        # here: tools/micro/_abl/lib_noswz.so
    python tools/variant_lib.py run noswz -- python tools/conv1x1_table.py   # on the GPU box: swaps the library in for
                                                                              # the command, restores it afterwards
"""
import os, shutil, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
ABL = os.path.join(ROOT, "tools", "micro", "_abl")
LIB = os.path.join(ROOT, "deeplio_amd", "libdeeplio_hip.so")


def build(name, flags):
    from deeplio_amd import build as B
    from deeplio_amd._header import abi_hash
    os.makedirs(os.path.join(ABL, name), exist_ok=True)
    srcs = sorted(f for f in os.listdir(B.CSRC) if f.endswith(".hip"))

    def cc(f):
        o = os.path.join(ABL, name, f[:-4] + ".o")
        subprocess.check_call([B.HIPCC] + B.FLAGS + ["-DDLIO_HEADER_CRC=%du" % abi_hash()] + flags +
                              ["-c", os.path.join(B.CSRC, f), "-o", o])
        return o
    with ThreadPoolExecutor(max_workers=6) as ex:
        objs = list(ex.map(cc, srcs))
    subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(ABL, "lib_%s.so" % name)] + objs)
    shutil.rmtree(os.path.join(ABL, name))
    print("built", os.path.join(ABL, "lib_%s.so" % name))


def run(name, cmd):
    keep = LIB + ".keep"
    shutil.copy(LIB, keep)
    try:
        shutil.copy(os.path.join(ABL, "lib_%s.so" % name), LIB)
        return subprocess.call(cmd)
    finally:
        shutil.move(keep, LIB)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build(sys.argv[2], sys.argv[3:])
    else:
        sys.exit(run(sys.argv[2], sys.argv[sys.argv.index("--") + 1:]))
