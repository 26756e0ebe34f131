"""Builds libsar_hip.so (the HIP library behind include/sar.h) in-tree with hipcc for gfx950.

    python -m strange_attractor_renderer_amd.build [--force]

The build also audits the device code: the iterate kernel must not contain a fused multiply-add
(v_fma_f64 / v_fmac_f64) — a single contraction changes the chaotic trajectories and breaks parity
with the reference (which Rust/LLVM never contracts).
"""
from __future__ import annotations

import hashlib
import os
import re
import shutil
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
OUT = os.path.join(PKG, "libsar_hip.so")
BUILD_DIR = os.path.join(os.path.dirname(PKG), "build", "sar_hip")
VARIANT_DIR = os.path.join(os.path.dirname(PKG), "build", "variants")   # A/B and test builds (SAR_LIBRARY=...), git-ignored
HOOKS_OUT = os.path.join(os.path.dirname(PKG), "tests", "hooks", "libsar_hip_hooks.so")   # product objects + sar_test_hooks.cpp
HOOKS_SOURCE = "sar_test_hooks.cpp"
SOURCES = ["sar_host.cpp", "sar_export.cpp", "sar_plan.cpp", "sar_render.cpp", "sar_runtime.cpp", "sar_batch.cpp", "sar_exchange.cpp", "sar_multi.cpp", "sar_iterate.hip", "sar_accumulate.hip",
           "sar_image.hip"]
HEADERS = [HOOKS_SOURCE, os.path.join("..", "..", "include", "sar_test_hooks.h"), "sar_internal.hpp", "sar_launch.hpp", "sar_device.hpp", "sar_runtime_impl.hpp", "sar_plan.hpp", os.path.join("..", "..", "include", "sar.h")]
ARCH = "gfx950"
FOLD_FUSED_OPS = 6   # v_fma_f64 in k_fold_resolve: the sqrt + div expansions of color_transform, nothing else

FLAGS = [
    f"--offload-arch={ARCH}", "-O3", "-std=c++17",
    "-ffp-contract=off",          # mandatory for bit parity (host AND device)
    "-fno-fast-math",
    "-fPIC", "-Wall", "-Wextra", "-Wno-unused-parameter", "-Wno-unused-value",
]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found")


def _extra_flags() -> list[str]:
    return os.environ.get("SAR_EXTRA_FLAGS", "").split() + os.environ.get("SAR_KERNEL_FLAGS", "").split()


def source_id(csrc: str | None = None, extra_flags: list[str] | None = None) -> str:
    """What a library is built FROM, as 16 hex digits: SHA-256 over every source and header of the library (csrc/*, include/sar.h;
    names and contents) and the compiler flags. The build embeds it (sar_build_id()); the loader recomputes it from the tree and
    refuses a binary built from other sources (_abi.load_library)."""
    csrc = csrc or CSRC
    h = hashlib.sha256()
    for name in sorted(SOURCES + HEADERS, key=os.path.basename):
        h.update(os.path.basename(name).encode() + b"\0")
        h.update(open(os.path.join(csrc, name), "rb").read())
        h.update(b"\0")
    h.update(" ".join(FLAGS + (extra_flags if extra_flags is not None else _extra_flags())).encode())
    return h.hexdigest()[:16]


def library_id(path: str) -> str | None:
    """The id embedded in a built library file (without loading it), or None."""
    try:
        data = open(path, "rb").read()
    except OSError:
        return None
    m = re.search(rb"SAR_BUILD_ID=([0-9a-f]{16})", data)
    return m.group(1).decode() if m else None


def _stale() -> bool:
    return library_id(OUT) != source_id() or library_id(HOOKS_OUT) != source_id()


def audit_no_fma(asm_paths) -> dict:
    """Counts fused fp64 ops per kernel in the device assembly; the kernels that run the map must have none."""
    text = "\n".join(open(p).read() for p in asm_paths)
    counts = {}
    # kernels are delimited by "<name>:" labels ... ".end_amdhsa_kernel"/"s_endpgm"
    # each function body runs from its "<name>:" label to the matching ".Lfunc_end<N>:" label
    for m in re.finditer(r"^(_ZN3sar\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, flags=re.S | re.M):
        name, body = m.group(1), m.group(2)
        counts[name] = len(re.findall(r"\bv_(fma|fmac|mad)_f64\b", body))
    bad = {k: v for k, v in counts.items() if any(t in k for t in ("k_iterate", "k_extent", "k_warmup")) and v}
    if bad:
        raise RuntimeError(f"fused fp64 ops found in the iterate kernel: {bad}")
    if not any("k_iterate" in k for k in counts):
        raise RuntimeError("audit could not find k_iterate in the device assembly")
    # k_fold_resolve replays next_point / screen_space from the checkpoints for the bit-exact `steps` payload, next to a
    # sqrt and a division whose correctly-rounded expansions legitimately use fused ops: exactly FOLD_FUSED_OPS of them
    # (sqrt 4, div 2... as emitted by ROCm 7.2's device libs). One more means the replay was contracted.
    fold = [v for k, v in counts.items() if "k_fold_resolve" in k]   # the single-frame kernel and its batched twin
    if fold != [FOLD_FUSED_OPS] * 2:
        raise RuntimeError(f"k_fold_resolve holds {fold} fused fp64 ops, expected [{FOLD_FUSED_OPS}] (sqrt/div expansion only): "
                           "either the payload replay was contracted or the device libs changed — inspect the assembly")
    return counts


LAST_BUILD = {"action": None, "id": None}   # what the last build_library() call did: "built" / "reused" and the id


def build_library(force: bool = False, verbose: bool = False, out: str | None = None, build_dir: str | None = None) -> str:
    """out / build_dir: build a VARIANT of the library somewhere else (with SAR_EXTRA_FLAGS / SAR_KERNEL_FLAGS set) for
    A/B timing or test builds; load it through the SAR_LIBRARY environment variable. The product is the default: it is
    rebuilt whenever the id embedded in the binary is not the id of the sources (contents, not mtimes)."""
    sid = source_id()
    if out is None and not force and not _stale():
        LAST_BUILD.update(action="reused", id=sid)
        return OUT
    OUT_ = out or OUT
    BUILD_DIR_ = build_dir or BUILD_DIR
    os.makedirs(BUILD_DIR_, exist_ok=True)
    hipcc = _hipcc()
    objs = []
    for s in SOURCES:
        obj = os.path.join(BUILD_DIR_, s + ".o")
        # SAR_EXTRA_FLAGS: extra -D flags for timing experiments (e.g. -DSAR_EXPERIMENT_...); never set by the product
        cmd = [hipcc, *FLAGS, *os.environ.get("SAR_EXTRA_FLAGS", "").split(), f'-DSAR_BUILD_ID="{sid}"', "-c", os.path.join(CSRC, s), "-o", obj]
        if s.endswith(".hip"):  # SAR_KERNEL_FLAGS: device-compiler flags for experiments (e.g. -mllvm options)
            cmd += ["-save-temps=obj", *os.environ.get("SAR_KERNEL_FLAGS", "").split()]
        if verbose:
            print(" ".join(cmd))
        subprocess.run(cmd, check=True, cwd=BUILD_DIR_)
        objs.append(obj)
    asm = [os.path.join(BUILD_DIR_, f"{s[:-4]}-hip-amdgcn-amd-amdhsa-{ARCH}.s") for s in SOURCES if s.endswith(".hip")]
    counts = audit_no_fma(asm)
    if verbose:
        print("fused-fp64 audit:", {k[:60]: v for k, v in counts.items()})
    tmp = OUT_ + ".tmp"
    subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, "-lz", "-lpthread", "-o", tmp], check=True)  # zlib: PNG export
    os.replace(tmp, OUT_)
    # the hooks build of the test-suite: the SAME object files plus the one that defines sar_runtime_set_test_option
    hooks_out = HOOKS_OUT if out is None else OUT_[:-3] + "_hooks.so"
    os.makedirs(os.path.dirname(hooks_out), exist_ok=True)
    hobj = os.path.join(BUILD_DIR_, HOOKS_SOURCE + ".o")
    subprocess.run([hipcc, *FLAGS, *os.environ.get("SAR_EXTRA_FLAGS", "").split(), "-c", os.path.join(CSRC, HOOKS_SOURCE), "-o", hobj], check=True, cwd=BUILD_DIR_)
    subprocess.run([hipcc, f"--offload-arch={ARCH}", "-shared", "-fPIC", *objs, hobj, "-lz", "-lpthread", "-o", hooks_out + ".tmp"], check=True)
    os.replace(hooks_out + ".tmp", hooks_out)
    LAST_BUILD.update(action="built", id=sid)
    return OUT_


if __name__ == "__main__":
    if "--variant" in sys.argv:  # python -m ...build --variant NAME  (flags from SAR_EXTRA_FLAGS / SAR_KERNEL_FLAGS)
        name = sys.argv[sys.argv.index("--variant") + 1]
        import tempfile
        scratch = os.environ.get("SAR_VARIANT_BUILD_DIR") or os.path.join(tempfile.gettempdir(), "sar_build")
        os.makedirs(VARIANT_DIR, exist_ok=True)   # variants never sit next to the product
        print(build_library(force=True, verbose=True, out=os.path.join(VARIANT_DIR, f"libsar_hip_{name}.so"),
                            build_dir=os.path.join(scratch, f"sar_hip_{name}")))
    else:
        print(build_library(force="--force" in sys.argv, verbose=True))
        print(f"{LAST_BUILD['action']} {LAST_BUILD['id']}")
