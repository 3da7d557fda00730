#!/usr/bin/env python
"""one-screen summary of a bench.py JSON line: step time, kernel classes, roofline.   usage: tools/bench_summary.py gpurun_out/x_bench.json"""
import json
import sys

d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(f"{d['ms_per_step']:.3f} ms/step  {d['value']:.0f} {d['unit']}  ({d.get('launch_mode')})  mfma_peak_frac {d.get('mfma_peak_frac', 0):.4f}")
for k, v in d.get("kernel_classes", {}).items():
    print(f"  {k:52s} {v['ms_per_step']:.3f} ms  {v['tflops']:7.1f} TF/s  x{v['launches_per_step']:.0f}")
kt = d.get("kernel_timer")
if kt:
    print(f"  eager {kt['eager_ms_per_step']:.3f} ms/step, timed classes {kt['timed_classes_ms_per_step']:.3f}, gates {kt.get('gates')}, passes {kt.get('passes_run')}")
r = d.get("roofline")
if r:
    if r.get("valid", True) and r.get("achieved") is not None:
        print(f"  roofline: {r['kernel']}: {r['achieved']:.0f} TF/s = {r['frac']:.3f} (issued {r['frac_issued']:.3f}), traffic {r.get('traffic')}")
    else:
        print(f"  roofline INVALID: {r.get('invalid_because')}")
a = d.get("attention_roofline")
if a:
    print(f"  attention_roofline: algorithmic {a['algorithmic']:.0f} TF/s = {a['frac_algorithmic']:.3f}, issued {a['frac_issued']:.3f}, {a['ms_per_step']:.3f} ms/step")
c = d.get("cpu_baseline")
if c:
    print(f"  cpu_baseline: {c.get('value')} {c.get('unit')} on {c.get('cores')} threads")
