"""Build libssd_hip.so (the C-ABI hot-path library) for gfx950 with hipcc, in-tree.

Usage:  python ssd-object-detection_amd/build.py [--force]
Each .hip file is compiled to an object with its own flags (match.hip needs unfused IEEE
arithmetic), then linked into ssd-object-detection_amd/libssd_hip.so.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libssd_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"

COMMON = ["-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function",
          "-I" + os.path.join(HERE, "..", "include")]
PER_FILE = {
    # bit-exact IEEE arithmetic vs numpy: no FMA contraction.  -fno-honor-nans only drops the
    # sNaN-quieting v_max(x,x) in front of every fmax/fmin (inputs are finite by contract).
    "match.hip": ["-ffp-contract=off", "-fno-honor-nans"],
    "detect.hip": ["-ffp-contract=off"],
    # the resize restates cv2's float arithmetic product by product (HIP's __fmul_rn / __fadd_rn are plain * and +)
    "prep.hip": ["-ffp-contract=off"],
    # ReLU is fmaxf(x, 0): with NaNs honoured every one of them is preceded by the sNaN-quieting v_max(x, x) -- 1 800 vector
    # instructions over the convolution epilogues, all on finite data
    "conv.hip": ["-fno-honor-nans"],
}


def _newer(src, dst, extra=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(s) > t for s in (src,) + tuple(extra))


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = tuple(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")) + (
        os.path.join(HERE, "..", "include", "ssd_hip.h"), os.path.abspath(__file__))
    srcs = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    objs = []
    procs = []
    for f in srcs:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f[:-4] + ".o")
        objs.append(obj)
        if force or _newer(src, obj, headers):
            cmd = [HIPCC] + COMMON + PER_FILE.get(f, []) + os.environ.get("SSD_EXTRA_HIPCC_FLAGS", "").split() + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((f, subprocess.Popen(cmd)))
    failed = [f for f, p in procs if p.wait() != 0]
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(failed))
    if force or procs or _newer(objs[0], LIB, tuple(objs)):
        cmd = [HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
