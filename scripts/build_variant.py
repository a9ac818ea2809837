"""LAB: a second copy of the library with ONE source compiled under extra defines, for same-box A/Bs of a compile-time lab
knob:   python scripts/build_variant.py <name> <source.hip>[,<source.hip>...|ALL] -DKNOB[=v] ...   ->  ait_amd/libait_hip_<name>.so
A script selects it with  AIT_LAB_LIB=<name>  (scripts/_lab_lib.py; never read by the product)."""
import os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ait_amd import build as B

name, srcs, defs = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build()
srcs = B.sources() if srcs == "ALL" else srcs.split(",")          # (a header knob: every source that includes it, or ALL)
objs = []
for s_ in B.sources():
    if s_ in srcs:
        obj = os.path.join(B.OBJ, s_[:-4] + "." + name + ".o")
        subprocess.check_call([B.hipcc()] + B.COMMON + B.PER_FILE.get(s_, []) + defs + ["-c", os.path.join(B.CSRC, s_), "-o", obj],
                              stderr=subprocess.DEVNULL)
    else:
        obj = os.path.join(B.OBJ, s_[:-4] + ".o")
    objs.append(obj)
out = os.path.join(B.HERE, "libait_hip_%s.so" % name)
subprocess.check_call([B.hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out] + objs)
print(out)
