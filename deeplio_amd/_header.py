"""Tiny parser for include/deeplio_hip.h: prototype names and parameter counts
(used by the CPU test-suite to prove the ctypes table and the .so match the header)."""
import os
import re

HEADER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "include", "deeplio_hip.h")


def prototypes(path=HEADER):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    src = re.sub(r"#[^\n]*", " ", src)
    src = re.sub(r"typedef\s+struct\s+\w+\s*\{.*?\}\s*\w+\s*;", " ", src, flags=re.S)
    out = {}
    for m in re.finditer(r"([\w\*\s]+?)\b(dlio_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        name, args = m.group(2), m.group(3).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        out[name] = n
    return out


def _stripped(path=HEADER):
    src = open(path).read()
    src = re.sub(r"/\*.*?\*/", " ", src, flags=re.S)
    src = re.sub(r"//[^\n]*", " ", src)
    return " ".join(src.split())


def abi_hash(path=HEADER):
    """CRC-32 of the comment-stripped, whitespace-normalised header: what build.py compiles into
    the library (dlio_abi_hash) and what _lib._load() compares at import"""
    import zlib
    return zlib.crc32(_stripped(path).encode()) & 0xffffffff


def abi_version(path=HEADER):
    m = re.search(r"#define\s+DLIO_ABI_VERSION\s+(\d+)", open(path).read())
    return int(m.group(1))
