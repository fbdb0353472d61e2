"""Import the upstream reference (read-only at /root/reference) in THIS container only.

The reference needs Python >= 3.11 (PEP 646 star-subscript at bsi/bsi.py:411) and
`jaxtyping`; neither is available here.  This shim installs a stub `jaxtyping`
and compiles `bsi/bsi.py` from an in-memory, one-token rewrite
(`x[*(...)]` -> `x[(...)]`, semantically identical).  Nothing is written under
/root/reference and nothing from it is copied into this repository: the shim is
used only by the fixture generators (`tools/gen_golden.py`, `tools/gen_golden_algos.py`,
`tools/gen_golden_drivers.py`), which run in the build container; nothing under `tests/`,
`bench.py` or `__graft_entry__.py` imports it (the GPU box has no /root/reference).
"""
import importlib
import os
import re
import sys
import types

REF_ROOT = os.environ.get("BSI_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isfile(os.path.join(REF_ROOT, "bsi", "bsi.py"))


def load():
    """Return the reference's `bsi.bsi` module (and make `bsi.nn`, `bsi.models` importable)."""
    if "bsi.bsi" in sys.modules and getattr(sys.modules["bsi.bsi"], "__ref_shim__", False):
        return sys.modules["bsi.bsi"]
    if not available():
        raise ImportError(f"reference not found under {REF_ROOT}")

    jt = types.ModuleType("jaxtyping")

    class _Ann:
        def __class_getitem__(cls, item):
            return item[0] if isinstance(item, tuple) else item

    for name in ("Float", "Int", "UInt8", "Bool", "Shaped", "Array"):
        setattr(jt, name, _Ann)
    sys.modules.setdefault("jaxtyping", jt)

    pkg = types.ModuleType("bsi")
    pkg.__path__ = [os.path.join(REF_ROOT, "bsi")]
    sys.modules["bsi"] = pkg

    path = os.path.join(REF_ROOT, "bsi", "bsi.py")
    with open(path) as fh:
        src = fh.read()
    src, n = re.subn(r"x\[\*\(\(None,\) \* \(lambda_\.ndim - 1\)\)\]",
                     "x[(None,) * (lambda_.ndim - 1)]", src)
    assert n == 1, "reference source changed; shim rewrite did not apply"
    mod = types.ModuleType("bsi.bsi")
    mod.__file__ = path
    mod.__ref_shim__ = True
    sys.modules["bsi.bsi"] = mod
    exec(compile(src, path, "exec"), mod.__dict__)
    pkg.bsi = mod
    return mod


def load_all():
    """Return a namespace with the reference classes used for golden vectors."""
    b = load()
    ns = types.SimpleNamespace()
    ns.bsi = b
    ns.BSI = b.BSI
    ns.Discretization = b.Discretization
    ns.LogUniform = b.LogUniform
    ns.dit = importlib.import_module("bsi.models.dit")
    ns.vdm_unet = importlib.import_module("bsi.models.vdm_unet")
    ns.pos_emb = importlib.import_module("bsi.models.pos_emb")
    ns.nn = importlib.import_module("bsi.nn")
    return ns
