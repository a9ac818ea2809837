"""Builds ait_amd/libait_hip.so (the C-ABI HIP library of include/ait_hip.h) for gfx950.

    python -m ait_amd.build [--force] [-v]

hipcc cross-compiles without a GPU.  Objects are cached under ait_amd/csrc/_obj and rebuilt
when the source, a header or the flags change.  The .so stays in-tree (git-ignored) so that it
travels to the GPU box with the repository snapshot.
"""
import hashlib
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libait_hip.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")
ARCH = "gfx950"

# -fno-slp-vectorize: hipcc's SLP vectoriser packs adjacent scalar f32 operations into v_pk_fma_f32 / v_pk_add_f32.
# In the GEMM epilogue (bias add over an accumulator register pair, the bias broadcast from the HIGH half of a register
# pair through op_sel, right behind a v_mov_b32 into the LOW half of that pair) the packed form returned, on some
# launches only, wrong low results in lanes 48-63 -- 16-element row segments off by the difference of two bias
# values, always the same four accumulator registers (scripts/gemm_lab diffmap with LAB_SELF=1: a kernel against a
# second launch of itself; ROCm 7.2, gfx950).  Without packing every kernel is reproducible launch to launch;
# MI355X_MICROARCH.md lists packed f32 arithmetic beside MFMAs as an anti-lever anyway.
# Round 6 looked for the cause at ISA level (profiles/r06_packed_f32_hazard.txt).  The vectorised epilogue is
#     v_mov_b32 v116, v142 ; s_waitcnt vmcnt(1) ; v_mov_b32 v117, v138 ; v_pk_fma_f32 v[114:115], s[44:45], v[96:97], v[116:117] op_sel:[0,0,1]
# (v138: a bias value just loaded; v[96:97]: accumulators).  REFUTED: a VALU -> packed-VALU forwarding hazard on the half
# taken through op_sel -- scripts/pk_hazard.hip repeats exactly that pair back to back 6e9 times per launch with a stale
# value in the register: not one wrong lane, with or without wait states.  What the symptom still fits (the LAST lane
# quarter of a register that is the destination of an in-flight load holding its previous content): the load's return
# against the counted s_waitcnt among the epilogue's stores -- not proven; the flag stays, it costs nothing measurable.
COMMON = ["-O3", "-fno-slp-vectorize", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-I", INCLUDE,
          "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]
# integer/geometry kernels must reproduce the reference's fp32 operation sequence exactly
EXACT = ["-ffp-contract=off"]
PER_FILE = {
    "roi_align.hip": EXACT,
    "roi_align_nhwc.hip": EXACT,
    "nms.hip": EXACT,
    "boxes.hip": EXACT,
}


def hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found")


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stamp(src, flags):
    h = hashlib.sha256()
    h.update(" ".join(flags).encode())
    for f in [os.path.join(CSRC, src)] + sorted(
            os.path.join(d, x) for d in (CSRC, INCLUDE) for x in os.listdir(d) if x.endswith(".h")):
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()


def _compile(src, force, verbose):
    flags = COMMON + PER_FILE.get(src, [])
    obj = os.path.join(OBJ, src[:-4] + ".o")
    stamp_file = obj + ".stamp"
    stamp = _stamp(src, flags)
    if not force and os.path.exists(obj) and os.path.exists(stamp_file) and \
            open(stamp_file).read() == stamp:
        return obj, False
    cmd = [hipcc()] + flags + ["-c", os.path.join(CSRC, src), "-o", obj]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stdout))
    if verbose and r.stdout.strip():
        print(r.stdout)
    with open(stamp_file, "w") as fh:
        fh.write(stamp)
    return obj, True


def build(force: bool = False, verbose: bool = False) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    with ThreadPoolExecutor(max_workers=min(6, len(srcs))) as ex:
        res = list(ex.map(lambda s: _compile(s, force, verbose), srcs))
    objs = [o for o, _ in res]
    if any(changed for _, changed in res) or not os.path.exists(LIB) or force:
        cmd = [hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="-v" in sys.argv))
