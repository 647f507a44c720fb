#!/usr/bin/env python3
"""Build-time lint of the spread_patch32_kernel ISA (patch32_kernels.h issues its matrix instructions from inline
assembly, which the compiler's hazard recogniser does not see):

  1. no basic block that holds a v_mfma may touch an accumulator with anything but v_mfma — neither the accumulation
     registers (a[..], v_accvgpr_*) nor the vector registers that v_mfma instructions of the kernel use as destination;
  2. no scratch (spill) traffic inside such a block;
  3. reports scratch bytes per kernel (informative).

usage: lint_patch32_isa.py file.s   (from hipcc --cuda-device-only -S).  Exit code 1 on a violation.
"""
import re
import sys


def regs_of(tok):
    """register numbers named by an operand like v12, v[4:7], a[0:3], a7"""
    m = re.fullmatch(r"([va])\[(\d+):(\d+)\]", tok)
    if m:
        return m.group(1), range(int(m.group(2)), int(m.group(3)) + 1)
    m = re.fullmatch(r"([va])(\d+)", tok)
    if m:
        return m.group(1), range(int(m.group(2)), int(m.group(2)) + 1)
    return None, ()


def operands(line):
    body = line.split(";")[0].strip()
    if not body or body.endswith(":") or body.startswith("."):
        return None, []
    parts = body.split(None, 1)
    ops = [t.strip() for t in parts[1].split(",")] if len(parts) > 1 else []
    return parts[0], [re.sub(r"\s.*", "", o) for o in ops]


def main(path):
    text = open(path).read().split("\n")
    bad = 0
    kernels = [i for i, l in enumerate(text) if re.match(r"^_ZN5nufft21spread_patch32_kernel[^:]*:", l)]
    for k in kernels:
        end = next(i for i in range(k, len(text)) if "s_endpgm" in text[i])
        body = text[k:end + 1]
        name = text[k].split(":")[0]
        # vector registers used as MFMA destinations
        vacc = set()
        for l in body:
            op, ops = operands(l)
            if op and op.startswith("v_mfma") and ops:
                cls, rr = regs_of(ops[0])
                if cls == "v":
                    vacc.update(rr)
        # basic blocks
        blocks, cur = [], []
        for l in body:
            if re.match(r"^\.LBB\d+_\d+:", l):
                blocks.append(cur)
                cur = []
            cur.append(l)
            op, _ = operands(l)
            if op and (op.startswith("s_cbranch") or op.startswith("s_branch") or op == "s_endpgm"):
                blocks.append(cur)
                cur = []
        blocks.append(cur)
        nm = 0
        for b in blocks:
            if not any("v_mfma" in l for l in b):
                continue
            for l in b:
                op, ops = operands(l)
                if not op:
                    continue
                if op.startswith("v_mfma"):
                    nm += 1
                    continue
                if "scratch_" in op:
                    print(f"{name}: spill traffic inside a matrix block: {l.strip()}")
                    bad += 1
                if op.startswith("v_accvgpr"):
                    print(f"{name}: accumulation register touched inside a matrix block: {l.strip()}")
                    bad += 1
                    continue
                for o in ops:
                    cls, rr = regs_of(o)
                    if cls == "a" or (cls == "v" and vacc.intersection(rr)):
                        print(f"{name}: accumulator register {o} used by a non-matrix instruction inside a matrix block: {l.strip()}")
                        bad += 1
        scratch = sum(1 for l in body if "scratch_" in l)
        print(f"{name}: {nm} matrix instructions in {sum(1 for b in blocks if any('v_mfma' in l for l in b))} blocks, "
              f"{len(vacc)} vector-register accumulators registers, {scratch} scratch instructions: {'ok' if not bad else 'VIOLATIONS'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
