"""oracle/ref_import.py -- TEST INFRASTRUCTURE ONLY; works only where /root/reference exists.

Makes the UNMODIFIED reference importable in this container so that oracle/gen_golden.py
can run it and record golden vectors.  Nothing from the reference is copied:
we only (a) put its directories on sys.path, (b) register empty stand-in modules for
third-party packages the reference imports at module scope but never uses on this path
(easydict is the exception: a 10-line attribute-dict is supplied because config.py builds
its global `cfg` from it), (c) patch os.popen('stty size') which the reference calls at
import time, and (d) inject the compiled reference operators (oracle/build_ref.py) as
`model._C`.
"""
import os
import sys
import types

REF = os.environ.get("AIT_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "lib", "model", "system"))


class _EasyDict(dict):
    """Minimal attribute dict with easydict's recursive-conversion behaviour."""

    def __init__(self, d=None, **kw):
        super().__init__()
        for k, v in dict(d or {}, **kw).items():
            setattr(self, k, v)

    def __setattr__(self, k, v):
        if isinstance(v, dict) and not isinstance(v, _EasyDict):
            v = _EasyDict(v)
        super().__setitem__(k, v)

    __setitem__ = __setattr__

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)


_done = False


def setup(with_native: bool = True):
    """Idempotently prepare sys.path / sys.modules for `import model...`."""
    global _done
    if _done:
        return
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF)
    for p in (REF, os.path.join(REF, "lib")):
        if p not in sys.path:
            sys.path.insert(0, p)

    def stub(name, **attrs):
        if name in sys.modules:
            return sys.modules[name]
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m
        return m

    stub("easydict", EasyDict=_EasyDict)
    stub("cv2")
    tv = stub("torchvision")
    tv.models = stub("torchvision.models")
    stub("termcolor", colored=lambda s, *a, **k: s, cprint=lambda *a, **k: None)

    real_popen = os.popen

    class _Fake:
        def read(self):
            return "24 80"

    def popen(cmd, *a, **k):
        if isinstance(cmd, str) and cmd.startswith("stty"):
            return _Fake()
        return real_popen(cmd, *a, **k)

    os.popen = popen

    if with_native:
        from . import build_ref
        import model  # the reference package (lib/model/__init__.py)
        model._C = build_ref.build()
        sys.modules["model._C"] = model._C
    _done = True


def reference_cfg():
    setup()
    from model.utils.config import cfg
    return cfg
