#!/usr/bin/env python3
"""Lint for kernels that issue loads from inline assembly (the weight-ring readers and the decoders' B-operand loads).

The compiler does not know that the result of an `asm volatile("global_load_dwordx4 ..." / "ds_read_b128 ...")` arrives later: it
treats the destination registers as written at the asm statement.  The kernels are correct only if NOTHING reads or overwrites those
registers before a wait that covers the load -- which the source arranges, but which a register copy inserted by the compiler (a
phi move at a loop edge, a re-materialisation) would silently break: the copy would read the registers before the data lands.
This tool checks the COMPILED code:

    python tools/asm_load_lint.py            # compiles the listed product files for gfx950 (-S) and scans the listed kernels

For every kernel it walks the instructions in program order (and once more around every loop, starting from the state at the
back edge) with two in-order queues, one per counter:
  vmcnt   every vector-memory load (compiler-issued ones and LDS-DMA included), lgkmcnt   every LDS read and scalar load;
`s_waitcnt vmcnt(N)` / `lgkmcnt(N)` retires all but the N youngest.  An instruction that names a register of a still-pending
ASM load, as a source or as a destination, is reported.  Scalar loads return out of order, so an lgkmcnt(N > 0) wait while a scalar
load is pending is reported too.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "point-cloud-compression_amd", "csrc")
# file -> kernel-name prefixes (mangled) to scan
TARGETS = {
    "encoder_fused_h2.hip": ["_Z23sa_pn_forward_h2_kernel"],
    "decoder_h2.hip": ["_Z18dec_main_h2_kernel"],
    "decoder.hip": ["_Z15dec_main_kernelILb1E"],
    # the planes kernels load their B operand from inline assembly into three rotating register sets.  Covered: the two-chunks-per-k-step
    # forms (MB = 8) in both arithmetics, the f16x2 chains and the wide bf16x3 chain.  NOT covered: the one-chunk-per-k-step forms
    # (planes_gemm_kernel<P, 4, ..>, planes_chain4_kernel<3, 1, ..>): their three-k-step trips have early-outs whose backward branches
    # this linear walker cannot tell from a full trip (it reports the set loaded by the trip's first k-step as pending at the header);
    # the call-to-call reproducibility test of tests/test_families.py runs those kernels on recycled memory instead.
    "planes.hip": ["_Z18planes_gemm_kernelILi3ELi8", "_Z18planes_gemm_kernelILi2ELi8", "_Z20planes_chain4_kernelILi2", "_Z20planes_chain4_kernelILi3ELi2"],
}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-honor-nans", "-I", os.path.join(ROOT, "include"),
         "-S", "--cuda-device-only"]

REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs(text):
    out = set()
    for a, b, c in REG.findall(text):
        if c:
            out.add(int(c))
        else:
            out.update(range(int(a), int(b) + 1))
    return out


def kernels(path, prefixes):
    lines = open(path).read().split("\n")
    i = 0
    while i < len(lines):
        l = lines[i]
        if "; @" in l and any(l.startswith(p) for p in prefixes):
            name = l.split(":")[0]
            j = i + 1
            while "s_endpgm" not in lines[j]:
                j += 1
            yield name, lines[i + 1:j]
            i = j
        i += 1


def scan(name, body):
    ins, labels = [], {}
    inasm = False
    for l in body:
        s = l.split(";")[0].strip() if "ASMSTART" not in l and "ASMEND" not in l else l.strip()
        if "ASMSTART" in l:
            inasm = True
            continue
        if "ASMEND" in l:
            inasm = False
            continue
        if not s or s.startswith("."):
            if s.endswith(":"):
                labels[s[:-1]] = len(ins)
            continue
        if s.endswith(":"):
            labels[s[:-1]] = len(ins)
            continue
        ins.append((s, inasm))
    problems = []
    # Several backward branches to ONE label: a loop whose trip runs up to three k-steps ("if (t + 1 < KT) kstep(t + 1)") is laid out with
    # its latch in front of the header, and every early-out of the trip jumps back to that latch -- where the trip-count test then leaves
    # the loop (t + 1 >= KT implies t + 3 >= KT).  Only the LAST backward branch to a label is a full trip that re-enters the body, so
    # only that edge is walked around the loop (assumption of this tool; it holds for the rotating-register loops it is pointed at).
    last_back = {}
    for i, (s_, _) in enumerate(ins):
        op_ = s_.split()[0]
        if op_.startswith("s_cbranch") or op_ == "s_branch":
            t_ = s_[len(op_):].strip()
            if t_ in labels and labels[t_] <= i:
                last_back[t_] = i

    def walk(start, end, vm, lg, seen_edges):
        i = start
        while i < end:
            s, in_asm = ins[i]
            op = s.split()[0]
            ops = s[len(op):]
            first = ops.split(",")[0] if ops.strip() else ""
            touched = regs(ops)
            pend = set().union(*[r for r, a in vm if a], *[r for r, a in lg if a]) if (vm or lg) else set()
            is_load = op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load"))
            is_lds = op.startswith("ds_read") or op.startswith("ds_load")
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", s)
                if m:
                    n = int(m.group(1))
                    vm[:] = vm[len(vm) - n:] if n else []
                m = re.search(r"lgkmcnt\((\d+)\)", s)
                if m:
                    n = int(m.group(1))
                    if n and any(r is None for r, _ in lg):
                        problems.append((i, s, "counted lgkmcnt while a scalar load is pending"))
                    lg[:] = lg[len(lg) - n:] if n else []
            else:
                addr_ok = set()
                if (is_load or is_lds) and in_asm:
                    addr_ok = regs(first)            # a load may overwrite its own (older, pending) destination only if the source means it
                hit = (touched - addr_ok) & pend if not ((is_load or is_lds) and in_asm) else (regs(ops[len(first):]) & pend)
                if hit:
                    problems.append((i, s, "touches v%s while an asm load into it is pending" % sorted(hit)[:4]))
                if is_load:
                    dest = regs(first) if "_lds_" not in op else set()
                    vm.append((dest, in_asm))
                elif is_lds:
                    lg.append((regs(first), in_asm))
                elif op.startswith("s_load") or op.startswith("s_buffer_load"):
                    lg.append((None, False))
                elif op.startswith(("ds_write", "ds_store", "ds_add", "ds_max", "ds_min", "ds_bpermute", "ds_swizzle", "ds_permute")):
                    lg.append((set(), False))
            if op.startswith("s_cbranch") or op == "s_branch":
                tgt = ops.strip()
                if tgt in labels and labels[tgt] <= i and last_back.get(tgt) == i and (labels[tgt], i) not in seen_edges:
                    seen_edges.add((labels[tgt], i))
                    walk(labels[tgt], i + 1, list(vm), list(lg), seen_edges)          # once around the loop with the state at the back edge
                elif op == "s_branch" and tgt in labels and labels[tgt] > i:
                    # an unconditional forward jump (a rotated loop: its continuation block sits in front of the header and is entered
                    # only through the back edge, which the branch above walks): program order continues at the target
                    i = min(labels[tgt], end)
                    continue
            i += 1

    walk(0, len(ins), [], [], set())
    n_asm = sum(1 for s, a in ins if a and s.split()[0].startswith(("global_load", "ds_read")))
    return n_asm, problems


def main():
    rc = 0
    with tempfile.TemporaryDirectory() as tmp:
        for f, prefixes in TARGETS.items():
            out = os.path.join(tmp, f + ".s")
            subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + [os.path.join(CSRC, f), "-o", out], check=True, stderr=subprocess.DEVNULL)
            for name, body in kernels(out, prefixes):
                n_asm, problems = scan(name, body)
                print(f"{f}: {name[:60]}  asm loads {n_asm}  problems {len(problems)}")
                for i, s, why in problems[:12]:
                    print(f"    #{i}: {s}   <- {why}")
                rc |= 1 if problems else 0
    return rc


if __name__ == "__main__":
    sys.exit(main())
