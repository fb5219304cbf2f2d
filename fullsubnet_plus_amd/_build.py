"""Build libfsnp_hip.so in-tree with hipcc (cross-compiles for gfx950 without a GPU).

Every csrc/*.hip is compiled to its own object (in parallel, cached by content hash under csrc/build/) and the objects are
linked into one shared library, so touching one kernel file costs one compile instead of eight."""
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(HERE, "libfsnp_hip.so")
SOURCES = ["fsnp_abi.hip", "fsnp_weights.hip", "fsnp_stft_abi.hip", "planner.cpp", "frontend.hip", "tcn.hip", "subband.hip", "lstm.hip", "lstm16.hip", "lstm_gru.hip", "lstm_coop.hip",
           "lstm_hp.hip", "lstm_hpw.hip", "lstm_generic.hip", "lstm_coopn.hip", "lstm_coopw.hip", "lstm_fbv.hip", "stft.hip", "box_probe.hip", "stages.hip"]
HEADERS = [os.path.join(CSRC, "fsnp_common.h"), os.path.join(CSRC, "lstm_common.h"), os.path.join(CSRC, "planner.h"), os.path.join(CSRC, "fsnp_handle.h"), os.path.join(CSRC, "weight_watch.h"),
           os.path.join(os.path.dirname(HERE), "include", "fsnp.h"), os.path.join(os.path.dirname(HERE), "include", "fsnp_debug.h")]
# -fno-slp-vectorize: the SLP pass pairs the LSTM kernel's per-tile VALU FMAs across tiles, which breaks the
# refill-in-place weight pipeline and makes hipcc drain vmcnt(0) + copy 48 registers every k-group
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC"]


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def _headers():
    return [h for h in HEADERS if os.path.exists(h)]


def _digest(paths):
    """Content hash (not mtimes: a gpurun snapshot or a fresh checkout resets them) of files + compile flags."""
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for p in paths:
        with open(p, "rb") as f:
            h.update(os.path.basename(p).encode() + b"\0" + f.read())
    return h.hexdigest()


def _read(path):
    try:
        with open(path) as f:
            return f.read().strip()
    except OSError:
        return ""


def source_digest():
    return _digest([os.path.join(CSRC, s) for s in _sources()] + _headers())


def is_stale():
    return not os.path.exists(LIB_PATH) or _read(LIB_PATH + ".stamp") != source_digest()


def _compile_one(hipcc, src, obj):
    res = subprocess.run([hipcc, *FLAGS, "-c", src, "-o", obj], capture_output=True, text=True)
    return src, res.returncode, res.stderr


def build(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -fPIC -c csrc/X.hip (in parallel) -> hipcc -shared -> libfsnp_hip.so."""
    if not force and not is_stale():
        return LIB_PATH
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    jobs, objs, stamps = [], [], {}
    for s in _sources():
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o").replace(".cpp", ".o"))
        objs.append(obj)
        stamps[obj] = _digest([src] + _headers())
        if force or not os.path.exists(obj) or _read(obj + ".stamp") != stamps[obj]:
            jobs.append((src, f"{obj}.tmp{os.getpid()}", obj))
    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as ex:
        results = list(ex.map(lambda j: _compile_one(hipcc, j[0], j[1]), jobs))
    errors = [f"{src}:\n{err[-4000:]}" for src, rc, err in results if rc != 0]
    if errors:
        for _, tmp, _ in jobs:
            if os.path.exists(tmp):
                os.remove(tmp)
        raise RuntimeError("hipcc failed:\n" + "\n".join(errors))
    for _, tmp, obj in jobs:
        os.replace(tmp, obj)                     # several ranks may build at once: compile aside, rename atomically
        with open(obj + ".stamp", "w") as f:
            f.write(stamps[obj])
    if verbose:
        for src, _, err in results:
            if err:
                print(src, err)
    tmp = f"{LIB_PATH}.tmp{os.getpid()}"
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", tmp, *objs], capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("hipcc link failed:\n" + res.stderr[-4000:])
    os.replace(tmp, LIB_PATH)
    with open(LIB_PATH + ".stamp", "w") as f:
        f.write(source_digest())
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
