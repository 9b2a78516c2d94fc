"""Build libnis_hip.so (gfx950) in-tree with hipcc.  `python -m nf_amd.build [--force] [--diag]`.

The per-conditioner-width translation units (forward: 5, wide adjoint: 3) compile in parallel; objects go to nf_amd/csrc/build/,
the shared library to nf_amd/lib/libnis_hip.so (git-ignored, travels to the GPU box with the snapshot).

`--diag` additionally builds nf_amd/lib/libnis_hip_diag.so: the same library with -DNIS_DIAG, i.e. plus the measured dead ends kept for
A/B measurements only -- the persistent LDS-resident C2 kernel (flow_mega.hip, NIS_MEGA=1), the multi-chain launch sets
(NIS_STREAMS=n) and the register-resident bijector of the matrix-core apply kernel (flow_mfma_reg.h, NIS_REGBIJ=1).  Select it
with NIS_LIB=<path>; the product library contains none of them.
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
        os.environ.get("NIS_EXTRA_FLAGS", "").split()          # diagnostics only (e.g. -DNIS_MEGA_STAMPS), never for the shipped library

LIB_DIAG = os.path.join(LIBDIR, "libnis_hip_diag.so")
API_UNITS = [("api", "api.hip", []), ("api_pack", "api_pack.hip", []), ("api_ingest", "api_ingest.hip", []), ("api_moments", "api_moments.hip", [])]
DIAG_UNITS = [("diag_api", "api.hip", ["-DNIS_DIAG"]), ("diag_api_pack", "api_pack.hip", ["-DNIS_DIAG"]), ("diag_flow_mega", "flow_mega.hip", ["-DNIS_DIAG"])] + \
             [(f"diag_flow_wm{w}", "flow_wm.hip", [f"-DNIS_WM={w}", "-DNIS_DIAG"]) for w in (16, 32, 64)]
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


def build(force=False, verbose=False, diag=False):
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
    diag = diag or os.path.exists(LIB_DIAG)          # once built, the diagnostics library is kept in step with the product library
    if diag and (force or _stale(LIB_DIAG, deps + [LIB])):
        with ThreadPoolExecutor(max_workers=min(6, os.cpu_count() or 2)) as ex:
            dres = list(ex.map(_compile, DIAG_UNITS))
        swapped = {"api.o", "api_pack.o", "flow_wm16.o", "flow_wm32.o", "flow_wm64.o"}
        objs = [os.path.join(OBJ, n + ".o") for n, _, _ in UNITS if n + ".o" not in swapped] + [o for o, _ in dres]
        _link(LIB_DIAG, objs)
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, verbose=True, diag="--diag" in sys.argv)
    print(path)
