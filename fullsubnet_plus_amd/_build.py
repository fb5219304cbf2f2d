"""Build libfsnp_hip.so in-tree with hipcc (cross-compiles for gfx950 without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.path.join(HERE, "libfsnp_hip.so")
SOURCES = ["fsnp_abi.hip", "frontend.hip", "tcn.hip", "subband.hip", "lstm.hip", "lstm_coop.hip", "lstm_coopn.hip", "stft.hip"]
HEADERS = [os.path.join(CSRC, "fsnp_common.h"), os.path.join(CSRC, "lstm_common.h"), os.path.join(os.path.dirname(HERE), "include", "fsnp.h")]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def is_stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = [os.path.join(CSRC, s) for s in SOURCES] + HEADERS
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -shared -fPIC csrc/*.hip -> libfsnp_hip.so
    (-fno-slp-vectorize: the SLP pass pairs the LSTM kernel's per-tile VALU FMAs across tiles, which breaks
    the refill-in-place weight pipeline and makes hipcc drain vmcnt(0) + copy 48 registers every k-group)."""
    if not force and not is_stale():
        return LIB_PATH
    tmp = f"{LIB_PATH}.tmp{os.getpid()}"       # several ranks may find the library stale at once: build aside, rename atomically
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-shared", "-fPIC", "-o", tmp]
    cmd += [os.path.join(CSRC, s) for s in SOURCES]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc failed:\n" + res.stderr[-4000:])
    if verbose and res.stderr:
        print(res.stderr)
    os.replace(tmp, LIB_PATH)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
