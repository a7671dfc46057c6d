"""Builds libmrs_swarm.so (HIP kernels + C ABI) in-tree for gfx950 with hipcc.

hipcc cross-compiles without a GPU, so this also is the "does it build" check on the CPU-only container.
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmrs_swarm.so")
ARCH = "gfx950"

# (source, -ffp-contract) — the LITERAL kernel and all host/init arithmetic must not be FMA-contracted
UNITS = [
    ("step_kernel_literal.hip", "off"),
    ("step_kernel_fast.hip", "fast"),
    ("collide.hip", "off"),
    ("outputs.hip", "off"),
    # host side (no kernels): C ABI, single-GPU tick, sharded tick, the three transports
    ("host_api.hip", "off"),
    ("tick_single.hip", "off"),
    ("tick_sharded.hip", "off"),
    ("transport_rccl.hip", "off"),
    ("transport_local.hip", "off"),
    ("transport_peer.hip", "off"),
]
DEPS = ["step_device.inc", "collide_device.inc", "swarm_layout.h", "host_internal.h", "sharded_protocol.h", os.path.join("..", "..", "include", "mrs_swarm.h")]


def _hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: libmrs_swarm.so cannot be built (there is no CPU fallback)")
    return exe


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def build_library(force=False, verbose=False):
    hipcc = _hipcc()
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    deps = [os.path.join(CSRC, d) for d in DEPS]
    objs = []
    for src, contract in UNITS:
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + deps):
            cmd = [hipcc, "-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", f"-ffp-contract={contract}",
                   "-fno-fast-math", "-Wall", "-Wno-unused-function", "-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
        objs.append(o)
    if force or _stale(LIB, objs):
        cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force=True, verbose=True))
