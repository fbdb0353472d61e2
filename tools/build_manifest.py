#!/usr/bin/env python
"""Write a manifest of the native build (compiler, flags, sources and their hashes, the library's hash and exported symbols):
`python tools/build_manifest.py > profiles/rN/build_manifest.txt` after `__graft_entry__.build()`."""
import hashlib
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "bsi_amd", "csrc")
LIB = os.path.join(ROOT, "bsi_amd", "lib", "libbsi_hip.so")


def sha(path):
    h = hashlib.sha256()
    with open(path, "rb") as f:
        for blk in iter(lambda: f.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def main():
    mk = open(os.path.join(CSRC, "Makefile")).read()
    print("compiler:", subprocess.run(["/opt/rocm/bin/hipcc", "--version"], capture_output=True, text=True).stdout.strip().splitlines()[0:2])
    print("flags   :", re.search(r"^CXXFLAGS\s*=\s*(.*)$", mk, re.M).group(1).replace("$(ARCH)", "gfx950").replace(" $(if $(LAB),-DBSI_LAB)", ""))
    try:
        head = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
        dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "bsi_amd/csrc", "include"], capture_output=True, text=True).stdout.strip()
        print("tree    :", head, "(sources modified)" if dirty else "(sources as committed)")
    except OSError:
        pass
    srcs = re.search(r"^SRCS\s*=\s*(.*)$", mk, re.M).group(1).split()
    print(f"sources : {len(srcs)} translation units + headers")
    for f in sorted(srcs + [h for h in os.listdir(CSRC) if h.endswith(".h")]) + ["../../include/bsi_hip.h"]:
        p = os.path.normpath(os.path.join(CSRC, f))
        print(f"  {sha(p)[:16]}  {os.path.getsize(p):8d}  {os.path.relpath(p, ROOT)}")
    if not os.path.isfile(LIB):
        print("library : NOT BUILT")
        return 1
    print(f"library : {os.path.relpath(LIB, ROOT)}  {os.path.getsize(LIB)} bytes  sha256 {sha(LIB)}")
    import ctypes
    hdr = open(os.path.join(ROOT, "include", "bsi_hip.h")).read()
    names = sorted(set(re.findall(r"\b(bsi_[a-z0-9_]+)\s*\(", hdr)))
    lib = ctypes.CDLL(LIB)
    missing = [n for n in names if not hasattr(lib, n)]
    print(f"exports : {len(names) - len(missing)} of the {len(names)} entry points include/bsi_hip.h declares" + (f"; MISSING {missing}" if missing else ""))
    return 0


if __name__ == "__main__":
    sys.exit(main())
