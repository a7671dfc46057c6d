#!/usr/bin/env python3
"""Kernel time vs swarm size (run on the GPU box): exposes the per-wave latency floor and the saturated throughput."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sizes = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "64,4096,65536,100000,131072,262144,524288,1000000,2000000").split(",")]
extra = sys.argv[2:] 
for n in sizes:
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "300", "--warmup", "30", "--uavs", str(n), "--no-cpu-baseline"] + extra,
                       capture_output=True, text=True)
    try:
        d = json.loads(r.stdout.strip().splitlines()[-1])
        print(f"N {n:8d}  value {d['value']:.3e}  us/step {d['ms_per_step']*1e3:8.2f}  kernel_us {d['roofline']['kernel_avg_ms']*1e3:8.2f}  frac {d['roofline']['frac']:.3f}", flush=True)
    except Exception:
        print(n, "FAILED", r.stderr[-400:])
