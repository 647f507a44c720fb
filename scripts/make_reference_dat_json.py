#!/usr/bin/env python3
"""Extracts the numbers the reference itself published for this path into tests/golden/reference_dat.json.

Source: /root/reference/benchmark/CPU+AMDGPU/results.MI300A_adastra/NonuniformFFTs_256_*.dat — the output of
benchmark/CPU+AMDGPU/run_benchmarks.jl (protocol :39-90: N = 256^3, sigma = 1.5, HalfSupport(4),
BackwardsKaiserBessel, coordinates randn folded into the period, values randn; columns 4 and 5 are the
relative l2 errors of the plan under test against an m = 8, sigma = 2 plan on the same data, :62-75).
These error columns are the only reference-generated numbers in the checkout (no Julia runtime here), so
they are what the oracle and the HIP path are pinned against (tests/test_reference_dat.py).  The timing
columns are kept for bench.py's annotations.  Only numbers (data) are extracted — no reference source text.

Run in the build container (needs /root/reference):  python3 scripts/make_reference_dat_json.py
"""
import json
import os
import re

REF = "/root/reference/benchmark/CPU+AMDGPU/results.MI300A_adastra"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "reference_dat.json")

FILES = {
    "Float64_ROC_shared": "NonuniformFFTs_256_Float64_ROCBackend_shared_memory.dat",
    "Float64_ROC_global": "NonuniformFFTs_256_Float64_ROCBackend_global_memory.dat",
    "ComplexF64_ROC_shared": "NonuniformFFTs_256_ComplexF64_ROCBackend_shared_memory.dat",
    "ComplexF64_ROC_global": "NonuniformFFTs_256_ComplexF64_ROCBackend_global_memory.dat",
    "Float64_CPU": "NonuniformFFTs_256_Float64_CPU.dat",
    "Float64_CPU_atomics": "NonuniformFFTs_256_Float64_CPU_atomics.dat",
    "ComplexF64_CPU": "NonuniformFFTs_256_ComplexF64_CPU.dat",
    "ComplexF64_CPU_atomics": "NonuniformFFTs_256_ComplexF64_CPU_atomics.dat",
}


def parse(path):
    header, rows = {}, []
    for line in open(path):
        line = line.rstrip("\n")
        if line.startswith("#"):
            m = re.match(r"#\s+-\s+([^:]+):\s*(.*)", line)
            if m:
                header[m.group(1).strip()] = m.group(2).strip()
            continue
        f = line.split()
        if len(f) >= 5:
            rows.append({"Np": int(f[0]), "type1_s": float(f[1]), "type2_s": float(f[2]),
                         "err_type1": float(f[3]), "err_type2": float(f[4])})
    return header, rows


def main():
    out = {"source": "benchmark/CPU+AMDGPU/results.MI300A_adastra/*.dat (reference checkout)",
           "protocol": "benchmark/CPU+AMDGPU/run_benchmarks.jl:39-90", "sets": {}}
    for key, name in FILES.items():
        header, rows = parse(os.path.join(REF, name))
        out["sets"][key] = {"file": name, "header": header, "rows": rows}
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    with open(OUT, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", OUT, {k: len(v["rows"]) for k, v in out["sets"].items()})


if __name__ == "__main__":
    main()
