"""Knob sweep of the persistent kernel on the C5 scene (all-features variant, two shade queues).  usage: python tools/gpu_c5_sweep.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
base = dict(os.environ, C5_CHECK="0")
for env in ({}, {"PRGPU_PP_SLOTS": "1024"}, {"PRGPU_PP_SLOTS": "768"}, {"PRGPU_PP_SHADE_MIN": "32"}, {"PRGPU_PP_SHADE_PARTIAL": "32"}, {"PRGPU_PP_REFILL": "32"},
            {"PRGPU_PP_REFILL": "56"}, {"PRGPU_PP_SLOTS": "1024", "PRGPU_PP_SHADE_PARTIAL": "32"}, {"PRGPU_PP_PARTIAL_ACT": "32"}, {"PRGPU_PP_BLOCKS_PER_CU": "2", "PRGPU_PP_SLOTS": "1024"}):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gpu_c5.py"), "12"], env=dict(base, **env), capture_output=True, text=True, timeout=300).stdout
    lines = [l for l in out.splitlines() if "Msamples/s" in l or "instrumented" in l]
    print(env or "default", "|", " | ".join(l.strip() for l in lines), flush=True)
