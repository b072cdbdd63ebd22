"""Build libmidihip.so (gfx950) in-tree with hipcc.  `python -m midi_model_amd.build` or `build()`.

The shared object lands next to this file so it travels with the repo snapshot to the GPU box;
objects go to midi-model_amd/_build/.  Sources are recompiled only when newer than their object.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "_build")
LIB = os.path.join(HERE, "libmidihip.so")
# the A/B test library: the same sources with -DMH_AB_BUILDS (first-form attention kernels, the 128x128 bf16 GEMM, the
# ablation / timeline builds of the production GEMM).  Loaded only by tests and tools (lib.use_ab()), never by the package.
OBJ_AB = os.path.join(HERE, "_build_ab")
LIB_AB = os.path.join(HERE, "libmidihip_ab.so")
INCLUDE = os.path.join(os.path.dirname(HERE), "include")

SOURCES = ["api.cpp", "comm.cpp", "gemm.hip", "gemm_pp256.hip", "gemm_skinny.hip", "elementwise.hip", "loss_optim.hip", "attention_small.hip", "attention_mfma.hip", "attention_mfma3.hip", "augment.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result"]
# per-file additions.  attention_mfma3.hip: no SLP vectorisation -- hipcc otherwise packs adjacent fp32 multiplies of the dS / rescale
# arithmetic into v_pk_mul_f32, and a packed fp32 VALU instruction beside MFMAs costs the matrix pipe more than the issue slot it
# saves (MI355X_MICROARCH.md, "price of one filler beside MFMAs").  Same bits (IEEE multiplies either way); same-box A/B, interleaved
# (profiles/r05_attn_noslp_ab.txt): backward 689 -> 672 us at 16 x 2048, 1988 -> 1936 us at 16 x 4096, forward +-0 (its main loop
# had no packed instruction).
FILE_FLAGS = {"attention_mfma3.hip": ["-fno-slp-vectorize"]}


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP kernels cannot be built")
    return exe


def _newest_header() -> float:
    t = 0.0
    for d in (CSRC, INCLUDE):
        for f in os.listdir(d):
            if f.endswith(".h"):
                t = max(t, os.path.getmtime(os.path.join(d, f)))
    return t


def build(force: bool = False, verbose: bool = False, ab: bool = True) -> str:
    """production library (+ the A/B test library unless ab=False); returns the production library's path"""
    hipcc = _hipcc()
    hdr_t = _newest_header()
    jobs = []
    variants = [(OBJ, LIB, [])] + ([(OBJ_AB, LIB_AB, ["-DMH_AB_BUILDS"])] if ab else [])
    for objdir, _, extra in variants:
        os.makedirs(objdir, exist_ok=True)
        for src in SOURCES:
            s = os.path.join(CSRC, src)
            o = os.path.join(objdir, os.path.splitext(src)[0] + ".o")
            if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(s), hdr_t):
                jobs.append((s, o, extra))

    def run(job):
        s, o, extra = job
        cmd = [hipcc, *FLAGS, *FILE_FLAGS.get(os.path.basename(s), []), *extra, "-x", "hip", "-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {s}:\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return o

    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            list(ex.map(run, jobs))
    for objdir, lib, _ in variants:
        objs = [os.path.join(objdir, os.path.splitext(s)[0] + ".o") for s in SOURCES]
        if any(o.startswith(objdir + os.sep) for _, o, _ in jobs) or not os.path.exists(lib):
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib, *objs]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"link failed:\n{r.stderr}")
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
