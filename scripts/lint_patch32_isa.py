#!/usr/bin/env python3
"""Build-time lint of the spread_patch32_kernel ISA (patch32_kernels.h issues its matrix instructions from inline
assembly, which the compiler's hazard recogniser does not see):

  1. no basic block that holds a v_mfma may touch an accumulator with anything but v_mfma — neither the accumulation
     registers (a[..], v_accvgpr_*) nor the vector registers that v_mfma instructions of the kernel use as destination;
  2. no scratch (spill) traffic inside such a block;
  3. no instruction may touch a register that an inline-assembly ds_read has been issued into before the next
     s_waitcnt lgkmcnt(0) (linear scan over the code layout: the kernels wait at the end of the block that issued);
  4. reports scratch instructions per kernel (informative).

usage: lint_patch32_isa.py file.s [--reads-only]   (from hipcc --cuda-device-only -S).  Exit code 1 on a violation.
--reads-only: the kernels of the file issue their matrix instructions through the compiler's builtin (which pads and
orders them itself): only check 3 applies (spread_patch_kernel of patch_kernels.h: its operand reads are inline assembly).
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


def main(path, reads_only=False):
    text = open(path).read().split("\n")
    bad = 0
    pat = r"^_ZN5nufft19spread_patch_kernel[^:]*:" if reads_only else r"^_ZN5nufft21spread_patch32_kernel[^:]*:"
    kernels = [i for i, l in enumerate(text) if re.match(pat, l)]
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
            if reads_only or not any("v_mfma" in l for l in b):
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
        # 3. registers with an inline-assembly LDS read in flight: forward data flow over the control-flow graph
        #    (pending set = union over predecessors; s_waitcnt lgkmcnt(0) clears it)
        labels, cfg_blocks, cur, cur_label = {}, [], [], None
        def close(fall):
            nonlocal cur, cur_label
            cfg_blocks.append({"label": cur_label, "lines": cur, "succ": [], "fall": fall})
            cur, cur_label = [], None
        in_asm = False
        for l in body[1:]:
            t = l.strip()
            mlab = re.match(r"^(\.LBB\d+_\d+):", l)
            if mlab:
                if cur or cur_label is not None:
                    close(True)
                cur_label = mlab.group(1)
                continue
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            op, ops = operands(l)
            if not op:
                continue
            cur.append((op, ops, in_asm, t))
            if op.startswith("s_cbranch"):
                cfg_blocks_target = ops[0]
                close(True)
                cfg_blocks[-1]["succ"].append(cfg_blocks_target)
            elif op == "s_branch":
                tgt = ops[0]
                close(False)
                cfg_blocks[-1]["succ"].append(tgt)
            elif op == "s_endpgm":
                close(False)
        if cur or cur_label is not None:
            close(False)
        index = {b["label"]: i for i, b in enumerate(cfg_blocks) if b["label"]}
        succs = []
        for i, b in enumerate(cfg_blocks):
            ss = [index[t] for t in b["succ"] if t in index]
            if b["fall"] and i + 1 < len(cfg_blocks):
                ss.append(i + 1)
            succs.append(ss)

        def transfer(b, pend, report):
            nb = 0
            pend = set(pend)
            for op, ops, asm, t in b["lines"]:
                if op == "s_waitcnt" and "lgkmcnt(0)" in t:
                    pend.clear()
                    continue
                used = set()
                for o in ops:
                    cls, rr = regs_of(o)
                    if cls == "v":
                        used.update(rr)
                if asm and op.startswith("ds_read"):
                    cls, rr = regs_of(ops[0])
                    if report and pend.intersection(used - set(rr)):
                        print(f"{name}: LDS read uses a register still in flight: {t}")
                        nb += 1
                    pend.update(rr)
                    continue
                hit = pend.intersection(used)
                if report and hit:
                    print(f"{name}: v{sorted(hit)[0]} has an inline-assembly LDS read in flight: {t}")
                    nb += 1
            return pend, nb

        inp = [set() for _ in cfg_blocks]
        changed = True
        while changed:
            changed = False
            for i, b in enumerate(cfg_blocks):
                out, _ = transfer(b, inp[i], False)
                for j in succs[i]:
                    if not out.issubset(inp[j]):
                        inp[j] |= out
                        changed = True
        for i, b in enumerate(cfg_blocks):
            _, nb = transfer(b, inp[i], True)
            bad += nb
        scratch = sum(1 for l in body if "scratch_" in l)
        print(f"{name}: {nm} matrix instructions in {sum(1 for b in blocks if any('v_mfma' in l for l in b))} blocks, "
              f"{len(vacc)} vector-register accumulators registers, {scratch} scratch instructions: {'ok' if not bad else 'VIOLATIONS'}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1], "--reads-only" in sys.argv[2:]))
