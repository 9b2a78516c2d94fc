"""Build libnis_hip.so (gfx950) in-tree with hipcc.  `python -m nf_amd.build [--force]`.

The per-conditioner-width translation units (forward: 5, wide adjoint: 3) compile in parallel; objects go to nf_amd/csrc/build/,
the shared library to nf_amd/lib/libnis_hip.so (git-ignored, travels to the GPU box with the snapshot).  ONE library: the measured
dead ends of rounds 2-3 (persistent LDS-resident step, multi-chain launch sets, register-resident bijector, producer / consumer
waves) were deleted in round 4; their A/B numbers are in profiles/r03_*_ab.txt and DESIGN.md, their code in the git history.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libnis_hip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=off", "-Wall", "-Wno-unused-function"] + \
        os.environ.get("NIS_EXTRA_FLAGS", "").split()          # A/B builds only (e.g. -DNIS_MF_REL=0), never for the shipped library

API_UNITS = [("api", "api.hip", []), ("api_pack", "api_pack.hip", []), ("api_ingest", "api_ingest.hip", []), ("api_moments", "api_moments.hip", [])]
UNITS = API_UNITS + [("rambo", "rambo.hip", []), ("flow_bwd", "flow_bwd.hip", [])] + \
        [(f"flow_wm{w}", "flow_wm.hip", [f"-DNIS_WM={w}"]) for w in (4, 8, 16, 32, 64)] + \
        [(f"flow_bwd_wm{w}", "flow_bwd_wm.hip", [f"-DNIS_WM={w}"]) for w in (16, 32, 64)]


def _sources():
    out = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))]
    out.append(os.path.join(os.path.dirname(HERE), "include", "nis_hip.h"))
    return out


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")] + [os.path.join(os.path.dirname(HERE), "include", "nis_hip.h")]


_FORCE = False


def _compile(unit):
    """Compile one translation unit unless its object is newer than its source and every header (units share the headers, not
    each other's sources: editing one .hip file rebuilds one object)."""
    name, src, defs = unit
    obj = os.path.join(OBJ, name + ".o")
    if not _FORCE and not os.environ.get("NIS_EXTRA_FLAGS") and not _stale(obj, [os.path.join(CSRC, src)] + _headers()):
        return obj, ""
    cmd = [HIPCC] + FLAGS + defs + ["-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed for {name}:\n{' '.join(cmd)}\n{r.stdout}\n{r.stderr}")
    return obj, r.stderr


def _link(lib, objs):
    cmd = [HIPCC, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", lib] + objs
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")


def build(force=False, verbose=False):
    global _FORCE
    _FORCE = bool(force)
    os.makedirs(OBJ, exist_ok=True)
    os.makedirs(LIBDIR, exist_ok=True)
    deps = _sources()
    if force or _stale(LIB, deps):
        with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
            results = list(ex.map(_compile, UNITS))
        if verbose:
            for _, err in results:
                if err.strip():
                    print(err, file=sys.stderr)
        _link(LIB, [o for o, _ in results])
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True)
    print(path)
