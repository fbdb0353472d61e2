"""The evidence under profiles/ must describe the tree that is shipped: every source hash listed in the NEWEST
profiles/r*/build_manifest.txt (tools/build_manifest.py, written as the last step of a round) has to match the file in the tree.
A kernel edited after the manifest (and after the profiles and the GPU suite it stands for) fails here."""
import glob
import hashlib
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_newest_build_manifest_matches_the_tree():
    manifests = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "build_manifest.txt")),
                       key=lambda p: int(re.search(r"r(\d+)", os.path.basename(os.path.dirname(p))).group(1)))
    assert manifests, "no profiles/r*/build_manifest.txt"
    newest = manifests[-1]
    rows = re.findall(r"^\s+([0-9a-f]{16})\s+(\d+)\s+(\S+)$", open(newest).read(), flags=re.M)
    assert len(rows) >= 20, f"{newest}: only {len(rows)} source rows"
    listed = {path for _, _, path in rows}
    stale = []
    for digest, size, path in rows:
        full = os.path.join(ROOT, path)
        assert os.path.isfile(full), f"{newest} lists {path}, which is not in the tree"
        data = open(full, "rb").read()
        if hashlib.sha256(data).hexdigest()[:16] != digest or len(data) != int(size):
            stale.append(path)
    assert not stale, f"{os.path.relpath(newest, ROOT)} does not describe the tree: {stale} changed after it was written " \
                      "(regenerate it -- and the profiles it stands for -- as the last step: python tools/build_manifest.py)"
    # and nothing the library is built from is missing from it
    csrc = os.path.join(ROOT, "bsi_amd", "csrc")
    built = {os.path.join("bsi_amd", "csrc", f) for f in os.listdir(csrc) if f.endswith((".hip", ".h"))} | {os.path.join("include", "bsi_hip.h")}
    mk = open(os.path.join(csrc, "Makefile")).read()
    srcs = set(re.search(r"^SRCS\s*=\s*(.*)$", mk, re.M).group(1).split())
    built = {p for p in built if p.endswith(".h") or os.path.basename(p) in srcs}
    assert built <= listed, f"sources missing from the manifest: {sorted(built - listed)}"
