"""Build recipe for libmobilequant_amd.so (HIP, gfx950 only): plain hipcc, in-tree output.

    python -m mobilequant_amd.build          # build if sources are newer than the library
    python -m mobilequant_amd.build --force
    python -m mobilequant_amd.build --experiments   # + the measured-negative kernel variants, into lib/experiments/ (MQ_LIB_PATH)

The library has a C ABI (include/mobilequant_amd.h) and no torch dependency; it is built in-tree at
mobilequant_amd/lib/ so it travels to the GPU box with the source snapshot.  hipcc cross-compiles
gfx950 without a GPU.  Flags that matter for bit-exact parity with the reference's fp32 arithmetic:
no fast-math, -ffp-contract=off, correctly rounded fp32 divide.
"""
from __future__ import annotations

import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIBDIR = os.path.join(PKG, "lib")
LIB = os.path.join(LIBDIR, "libmobilequant_amd.so")
SOURCES = ["mq_elementwise.hip", "mq_reduce.hip", "mq_gemm.hip", "mq_gemm_grouped.hip", "mq_gemv.hip", "mq_norm.hip", "mq_decode.hip", "mq_attention.hip", "mq_qmatmul.hip"]
# per-file additions: the attention kernel is VALU-bound and consumes every MFMA result with VALU instructions -- keep the MFMA
# results in VGPRs (no v_accvgpr_read per score element)
# mq_decode.hip: no SLP vectorisation -- a v_pk_mul_f32 names a register PAIR although op_sel reads one half; when the other half
# is the destination of a load still in flight, hipcc's waitcnt pass waits for it (vmcnt(0) = the whole weight stream) in front of
# the activation-image arithmetic
PER_FILE_FLAGS = {"mq_attention.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "mq_decode.hip": ["-fno-slp-vectorize"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ABLATE = (["-DMQ_GEMM_ABLATE"] if os.environ.get("MQ_GEMM_ABLATE") else []) + \
    ([f"-DMQ_PP_PRIO={os.environ['MQ_PP_PRIO']}"] if os.environ.get("MQ_PP_PRIO") else [])
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off", "-fno-fast-math", "-fhip-fp32-correctly-rounded-divide-sqrt",
    "-Wall", "-Wno-unused-function",
    "-I" + os.path.join(ROOT, "include"), "-I" + CSRC, *ABLATE,
]


def _newest_source() -> float:
    files = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(ROOT, "include", "mobilequant_amd.h"),
                                                               os.path.abspath(__file__)]
    return max(os.path.getmtime(f) for f in files)


def needs_build() -> bool:
    return (not os.path.exists(LIB)) or os.path.getmtime(LIB) < _newest_source()


def build(force: bool = False, verbose: bool = False, tag: str = "", extra_flags=(), only=None) -> str:
    """Compile every HIP source for gfx950 and link the shared library.  Returns its path.
    tag: build an experiment copy into lib/<tag>/ (A/B timing through MQ_LIB_PATH; never loaded by default);
    only: with a tag, recompile just these sources and link the main build's objects for the rest."""
    libdir = os.path.join(LIBDIR, tag) if tag else LIBDIR
    lib = os.path.join(libdir, "libmobilequant_amd.so")
    if not force and not tag and not needs_build():
        return LIB
    os.makedirs(libdir, exist_ok=True)
    objs = []
    procs = []
    for src in SOURCES:
        if tag and only is not None and src not in only:
            objs.append(os.path.join(LIBDIR, src.replace(".hip", ".o")))
            continue
        obj = os.path.join(libdir, src.replace(".hip", ".o"))
        cmd = [HIPCC, *FLAGS, *PER_FILE_FLAGS.get(src, ()), *extra_flags, "-c", os.path.join(CSRC, src), "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)))
        objs.append(obj)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out}")
        if verbose and out.strip():
            print(out)
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *objs, "-o", lib]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}")
    return lib


def build_probe(tag: str = "") -> str:
    """tools/mq_probe (standalone C++ checker / timer of the GEMM variants), bound to lib/<tag>/ by rpath."""
    libdir = os.path.join(LIBDIR, tag) if tag else LIBDIR
    probe = os.path.join(ROOT, "tools", "mq_probe" + ("_" + tag if tag else ""))
    rpath = "$ORIGIN/../mobilequant_amd/lib" + ("/" + tag if tag else "")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O2", "-w", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tools", "mq_probe.cpp"), "-L" + libdir, "-lmobilequant_amd", "-Wl,-rpath," + rpath,
                    "-Wl,--disable-new-dtags", "-o", probe], check=True)
    return probe


if __name__ == "__main__":
    tag = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--tag=")), "")
    # --experiments: also compile the measured-negative kernel variants (split-K residual tiles, per-wave W4 unpack) that the production
    # library leaves out; goes to lib/experiments/ unless a tag is given, never loaded by default (MQ_LIB_PATH selects it)
    exp = "--experiments" in sys.argv
    if exp and not tag:
        tag = "experiments"
    gen = []
    if exp:     # the experiment-only generated programs are not committed: write them next to the others for this build, remove them after
        import importlib.util
        spec = importlib.util.spec_from_file_location("gen_fr_asm", os.path.join(ROOT, "tools", "gen_fr_asm.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        for v in mod.EXPERIMENTAL:
            mod.main(variant=v)
            gen.append(os.path.join(CSRC, mod.VARIANTS[v][4]))
    try:
        path = build(force="--force" in sys.argv or exp, verbose=True, tag=tag, extra_flags=["-DMQ_BUILD_EXPERIMENTS"] if exp else ())
    finally:
        for f in gen:
            os.remove(f)
    print("built", path)
    if tag:
        print("built", build_probe(tag))
