"""oracle/digest.py -- TEST INFRASTRUCTURE ONLY.

Compact, size-bounded summaries of large tensors for golden fixtures ("strided
subsamples + full-tensor sum/abs-sum/L2 for big ones") and seeded input generation shared by
oracle/gen_golden.py and tests/.
"""
import numpy as np

FULL_MAX = 1 << 15     # tensors up to this many elements are stored whole
N_SAMPLE = 4096


def seeded(seed, shape, scale=1.0, dtype=np.float32):
    """Deterministic N(0, scale^2) array on the frozen MT19937 stream."""
    return (np.random.RandomState(seed).standard_normal(size=shape) * scale).astype(dtype)


def _np(t):
    if hasattr(t, "detach"):
        t = t.detach().cpu().numpy()
    return np.asarray(t)


def digest(t):
    """dict(shape, sum, abssum, l2, sample[, full]) in float64 / source dtype."""
    a = _np(t)
    flat = a.reshape(-1)
    d = {
        "shape": np.asarray(a.shape, np.int64),
        "sum": np.float64(flat.astype(np.float64).sum()),
        "abssum": np.float64(np.abs(flat.astype(np.float64)).sum()),
        "l2": np.float64(np.sqrt((flat.astype(np.float64) ** 2).sum())),
    }
    if flat.size <= FULL_MAX:
        d["full"] = a.copy()
    else:
        stride = max(1, flat.size // N_SAMPLE)
        d["sample"] = flat[::stride][:N_SAMPLE].copy()
    return d


def pack(prefix, t, out):
    for k, v in digest(t).items():
        out["%s/%s" % (prefix, k)] = v


def compare(prefix, t, golden, rtol, atol):
    """Return (ok, message).  Elementwise on full/sample; aggregates scaled by abssum."""
    d = digest(t)
    g = {k[len(prefix) + 1:]: golden[k] for k in golden.files if k.startswith(prefix + "/")}
    if not g:
        return False, "no golden entry %r" % prefix
    if list(d["shape"]) != list(g["shape"]):
        return False, "%s: shape %s vs golden %s" % (prefix, d["shape"], g["shape"])
    key = "full" if "full" in g else "sample"
    a, b = np.asarray(d[key], np.float64), np.asarray(g[key], np.float64)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    if not (err <= tol).all():
        i = int(np.argmax(err - tol))
        return False, "%s[%s]: max viol at %d: got %.9g want %.9g (rtol %g atol %g)" % (
            prefix, key, i, a.reshape(-1)[i], b.reshape(-1)[i], rtol, atol)
    n = max(1, int(np.prod(d["shape"])))
    # aggregate checks: a sum of n terms each within (atol + rtol*|x|)
    budget = atol * n + rtol * float(g["abssum"])
    if abs(float(d["sum"]) - float(g["sum"])) > budget:
        return False, "%s: sum %.9g vs %.9g" % (prefix, d["sum"], g["sum"])
    if abs(float(d["l2"]) - float(g["l2"])) > atol * np.sqrt(n) + rtol * float(g["l2"]):
        return False, "%s: l2 %.9g vs %.9g" % (prefix, d["l2"], g["l2"])
    return True, "ok"
