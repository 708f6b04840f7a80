#!/usr/bin/env python3
"""Prints the headline numbers of one bench.py JSON line read from stdin: tools/bench_line.py <label>"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(sys.argv[1] if len(sys.argv) > 1 else "", round(d["value"] / 1e6, 3), "M steps/s", round(d["roofline"]["kernel_ms_per_launch"], 2), "ms/launch",
      *(["| fast mode", round(d["fp32_fast_mode"]["value"] / 1e6, 3), "M steps/s", round(d["fp32_fast_mode"]["kernel_ms_per_launch"], 2), "ms/launch"] if "fp32_fast_mode" in d else []))
