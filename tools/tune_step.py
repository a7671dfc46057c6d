#!/usr/bin/env python3
"""Kernel tuning sweep (run on the GPU box): builds variants of the step kernel with different -D knobs, runs bench.py
against each through MRS_SWARM_LIB and prints one line per variant.  Usage: python tools/tune_step.py [--workload ...]"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "mrs_multirotor_simulator_amd", "csrc")
OBJ = os.path.join(ROOT, "mrs_multirotor_simulator_amd", "build")

VARIANTS = {
    "default": "",
    "w1": "-DMRS_WAVES_PER_SIMD=1",
    "w3": "-DMRS_WAVES_PER_SIMD=3",
    "unroll1": "-DMRS_SU=1",
    "unroll2": "-DMRS_SU=2",
    "maxilp": "-mllvm -amdgpu-sched-strategy=max-ilp",
    "maxilp_w1": "-mllvm -amdgpu-sched-strategy=max-ilp -DMRS_WAVES_PER_SIMD=1",
    "memclause": "-mllvm -amdgpu-sched-strategy=max-memory-clause",
    "O2": "-O2",
    "nolicm": "-mllvm -disable-licm-promotion",
    "st_nt": "-DMRS_ST_AUX=2",
    "st_sc1": "-DMRS_ST_AUX=16",
    "st_sc01": "-DMRS_ST_AUX=17",
    "st_sc01nt": "-DMRS_ST_AUX=19",
    "ld_nt": "-DMRS_LD_AUX=2",
    "ldst_nt": "-DMRS_LD_AUX=2 -DMRS_ST_AUX=2",
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="actuator")
    ap.add_argument("--arith", default="fast,literal")
    ap.add_argument("--variants", default=",".join(VARIANTS))
    ap.add_argument("--steps", type=int, default=500)
    ap.add_argument("--uavs", type=int, default=100000)
    ap.add_argument("--extra", default="", help="extra -D flags applied to every variant")
    args = ap.parse_args()
    sys.path.insert(0, ROOT)
    from mrs_multirotor_simulator_amd import build
    build.build_library()
    out_dir = "/tmp/mrs_variants"
    os.makedirs(out_dir, exist_ok=True)
    for name in args.variants.split(","):
        if name.startswith("aux_"):  # aux_<load bits>_<store bits>: cache policy of the state accesses
            _, la, sa = name.split("_")
            VARIANTS[name] = f"-DMRS_LD_AUX={la} -DMRS_ST_AUX={sa}"
        flags = (VARIANTS[name] + " " + args.extra).split()
        objs = []
        for v, c in (("literal", "off"), ("fast", "fast")):
            o = os.path.join(out_dir, f"{name}_{v}.o")
            subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-fPIC", "-std=c++17", f"-ffp-contract={c}",
                                   "-fno-fast-math"] + flags + ["-c", os.path.join(CSRC, f"step_kernel_{v}.hip"), "-o", o])
            objs.append(o)
        lib = os.path.join(out_dir, f"libmrs_{name}.so")
        subprocess.check_call(["hipcc", "-shared", "-fPIC", "--offload-arch=gfx950", "-o", lib] + objs +
                              [os.path.join(OBJ, "collide.o"), os.path.join(OBJ, "outputs.o"), *[os.path.join(OBJ, o) for o in ("host_api.o", "tick_single.o", "tick_sharded.o", "transport_rccl.o", "transport_local.o", "transport_peer.o")]])
        for arith in args.arith.split(","):
            env = dict(os.environ, MRS_SWARM_LIB=lib)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(args.steps), "--warmup", "50",
                                "--arith", arith, "--workload", args.workload, "--uavs", str(args.uavs), "--no-cpu-baseline"],
                               env=env, capture_output=True, text=True)
            try:
                d = json.loads(r.stdout.strip().splitlines()[-1])
                print(f"{name:14s} {arith:8s} {args.workload:10s} value {d['value']:.3e}  us/step {d['ms_per_step'] * 1e3:7.2f}  "
                      f"kernel_us {d['roofline']['kernel_avg_ms'] * 1e3:7.2f}  frac {d['roofline']['frac']:.3f}", flush=True)
            except Exception:
                print(name, arith, "FAILED", r.stdout[-300:], r.stderr[-500:], flush=True)


if __name__ == "__main__":
    main()
