"""oracle/build_ref.py -- TEST INFRASTRUCTURE ONLY.

Compiles the reference's own CPU operators (RoIAlign forward, NMS) from the sources where
they lie under /root/reference/lib/model/csrc (vision.cpp:7-13, cpu/ROIAlign_cpu.cpp,
cpu/nms_cpu.cpp) into oracle/_ref/ref_C*.so.  Nothing is copied into this repository; the
output directory is git-ignored.  The only build concession is the force-included
oracle/ref_torch_compat.h (see its header) which restores a deprecated torch overload.

The resulting pybind module is used by oracle/gen_golden.py to produce tests/golden/*.npz
and by tests (when present) to cross-check oracle/native.c.  It is never imported by
ait_amd/.
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("AIT_REFERENCE_ROOT", "/root/reference")
CSRC = os.path.join(REF, "lib", "model", "csrc")
OUT = os.path.join(HERE, "_ref")


def available() -> bool:
    return os.path.isdir(CSRC)


def build(verbose: bool = False):
    """Build (or load the cached) reference `_C` module. Returns the module or None."""
    import importlib.util
    import glob

    os.makedirs(OUT, exist_ok=True)
    prebuilt = sorted(glob.glob(os.path.join(OUT, "ref_C*.so")))
    if not available():
        if not prebuilt:
            return None
        import torch  # noqa: F401  (registers libtorch symbols)
        spec = importlib.util.spec_from_file_location("ref_C", prebuilt[0])
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        return mod
    from torch.utils.cpp_extension import load

    return load(
        name="ref_C",
        sources=[
            os.path.join(CSRC, "vision.cpp"),
            os.path.join(CSRC, "cpu", "ROIAlign_cpu.cpp"),
            os.path.join(CSRC, "cpu", "nms_cpu.cpp"),
        ],
        extra_include_paths=[CSRC],
        extra_cflags=["-O2", "-include", os.path.join(HERE, "ref_torch_compat.h"), "-w"],
        build_directory=OUT,
        verbose=verbose,
    )


if __name__ == "__main__":
    m = build(verbose="-v" in sys.argv)
    print("reference _C:", m)
