"""In-tree build of libneko_hip.so (gfx950 only) with hipcc -- no hipify, no JIT cache.

    python -m neko_amd.build            # incremental
    python -m neko_amd.build --force

Objects go to neko_amd/csrc/build/, the library to neko_amd/csrc/libneko_hip.so (git-ignored,
travels to the GPU box with the snapshot).  hipcc cross-compiles without a GPU.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
BUILD = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libneko_hip.so")
SOURCES = ["gemm_bf16.hip", "gemm_glds.hip", "gemm_a16.hip", "gemm_b16.hip", "gemm_p16.hip", "gemv_bf16.hip", "layernorm.hip", "attention.hip", "attention_res.hip", "attention_stream.hip", "attention_decode.hip", "cross_entropy.hip", "elementwise.hip",
           "pack_embed.hip", "patch_embed.hip", "rows.hip", "segsum.hip", "dropout.hip", "neko_capi.hip"]
HEADERS = ["neko_common.h", "neko_kernels.h", os.path.join("..", "..", "include", "neko_hip.h")]
# headers only some sources include (a change rebuilds just those)
EXTRA_DEPS = {"gemm_glds.hip": ["gemm_epi.h"], "gemm_a16.hip": ["gemm_epi.h", "gemm_a16_loop.inc"],
              "gemm_b16.hip": ["gemm_epi.h", "gemm_b16_loop.inc"], "gemm_p16.hip": ["gemm_epi.h", "gemm_p16_loop.inc"]}
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-ffp-contract=on", "-Wno-unused-result"]
FLAGS += os.environ.get("NEKO_EXTRA_HIPCC_FLAGS", "").split()


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (set HIPCC)")


def _stale(target: str, deps) -> bool:
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile(src: str, force: bool) -> str:
    obj = os.path.join(BUILD, src.replace(".hip", ".o"))
    deps = [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS + EXTRA_DEPS.get(src, [])]
    if force or _stale(obj, deps):
        cmd = [_hipcc(), *FLAGS, "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
    return obj


def build(force: bool = False, jobs: int = 4, verbose: bool = True) -> str:
    global BUILD, LIB
    tag = os.environ.get("NEKO_BUILD_TAG")          # A/B builds: separate object dir and library name
    if tag:
        BUILD = os.path.join(CSRC, "build_" + tag)
        LIB = os.path.join(CSRC, f"libneko_hip_{tag}.so")
    os.makedirs(BUILD, exist_ok=True)
    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(lambda s: _compile(s, force), SOURCES))
    if force or _stale(LIB, objs):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", *objs, "-o", LIB]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    if verbose:
        print(f"[neko_amd.build] {LIB} ({os.path.getsize(LIB) / 1e6:.1f} MB)")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
