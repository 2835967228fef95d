"""Opcode histogram + issue-cost bound of a kernel's basic blocks, from the compiler's own assembly.

    hipcc --offload-arch=gfx950 -O3 ... -S --cuda-device-only lk.hip -o lk.s
    python tools/isa_histogram.py lk.s lk_circular_kernel [--cost profiles/r02_valu_issue_cost.txt] [--blocks a,b,c]

For every basic block of the kernel (labels `.LBB0_n:` and the fall-through markers `; %bb.n:`) it prints the number of
VALU / SALU / LDS / VMEM instructions and the block's VALU issue cost = sum over opcodes of count x measured cost
(SIMD cycles per wave64 instruction, from the rocprofv3 --pmc pass of tools/ubench/valu_rate.hip; opcodes that were
not measured are priced at the 4.4-cycle class and listed).  --blocks sums a chosen path (e.g. the hot path of the
Gauss-Newton iteration) and prints its histogram: that sum x the dynamic iteration count is the VALU-issue bound the
kernel's measured time is compared with.
"""
import argparse
import re
from collections import Counter, OrderedDict

DEFAULT_CLASS = 4.4


def load_costs(path):
    cost = {}
    if not path:
        return cost
    for line in open(path):
        m = re.match(r"\s*k_(\w+)\s+\d+\s+\d+\s+([\d.]+)\s+[\d.]+", line)
        if m:
            cost[m.group(1)] = float(m.group(2))
    return cost


def base_op(op):
    op = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    return op


def lookup(op, cost):
    b = base_op(op)[2:] if op.startswith("v_") else None
    if b is None:
        return None
    alias = {"dot2c_i32_i16": "dot2c_i32_i16", "readlane_b32": "readlane", "readfirstlane_b32": "readlane",
             "permlane32_swap_b32": "permlane32_swap", "permlane16_swap_b32": "permlane16_swap", "mov_b32": "mov",
             "lshrrev_b32": "lshlrev_b32", "mad_i32_i24": "mad_u32_u24", "mov_b64": "mov", "or_b32": "and_b32",
             "max_u32": "add_u32", "min_u32": "add_u32", "cvt_f64_f32": "add_f64", "add_u32": "add_u32"}
    if b.startswith("cmp_") or b.startswith("cmpx_"):
        b = "cmp_lt_f32"
    if "dpp" in op and b == "add_u32":
        b = "add_u32_dpp"
    if "dpp" in op and b == "mov_b32":
        b = "mov_b32_dpp"
    b = alias.get(b, b)
    return cost.get(b)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("asm")
    ap.add_argument("kernel")
    ap.add_argument("--cost", default=None)
    ap.add_argument("--blocks", default=None, help="comma-separated block names to sum (e.g. LBB0_35,bb.38,LBB0_39)")
    ap.add_argument("--floor", action="store_true",
                    help="price every opcode at the FLOOR of its class instead of its own measured cost: 2.46 for the class the "
                         "micro-benchmark measures below 3 cycles, 8.0 for the one it measures above 7, 4.0 (one wave64 pass of the "
                         "16-lane SIMD) for everything else incl. unmeasured opcodes -- a sum no issue schedule can beat, which is "
                         "what bench.py's valu_issue_frac divides by the measured cycles (round 6: <= 1 by construction)")
    args = ap.parse_args()
    cost = load_costs(args.cost)
    if args.floor:
        cost = {k: (2.46 if v < 3.0 else 8.0 if v > 7.0 else 4.0) for k, v in cost.items()}
        cost["cndmask_b32"] = 4.0  # (the 23-cycle row is an artefact of a back-to-back stream of itself, r02_lk_issue_bound.md)
        global DEFAULT_CLASS
        DEFAULT_CLASS = 4.0
    blocks = OrderedDict()
    cur, inside = None, False
    for line in open(args.asm):
        if not inside:
            if re.match(r"^_Z\w*%s\w*:" % re.escape(args.kernel), line):
                inside, cur = True, "entry"
                blocks[cur] = []
            continue
        if line.startswith(".Lfunc_end") or line.strip().startswith(".section"):
            break
        m = re.match(r"^\.(LBB\d+_\d+):", line) or re.match(r"^; %(bb\.\d+):", line)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            continue
        t = line.strip()
        if not t or t.startswith(";") or t.startswith("."):
            continue
        op = t.split()[0]
        if re.match(r"^[vs]_|^ds_|^global_|^buffer_|^flat_|^scratch_", op):
            ctrl = " dpp" if ("quad_perm" in t or "row_" in t) else ""
            blocks[cur].append(op + ("_dpp" if ctrl and not op.endswith("_dpp") else ""))
    unknown = Counter()

    def summarise(ops):
        valu = [o for o in ops if o.startswith("v_")]
        c = 0.0
        for o in valu:
            k = lookup(o, cost)
            if k is None:
                unknown[base_op(o)] += 1
                k = DEFAULT_CLASS
            c += k
        return dict(valu=len(valu), salu=sum(o.startswith("s_") and not o.startswith("s_nop") and not o.startswith("s_waitcnt") for o in ops),
                    nops=sum(o.startswith("s_nop") for o in ops), lds=sum(o.startswith("ds_") for o in ops),
                    vmem=sum(o.split("_")[0] in ("global", "buffer", "flat", "scratch") for o in ops), cycles=c)

    print("%-10s %5s %5s %5s %4s %5s %9s" % ("block", "VALU", "SALU", "s_nop", "LDS", "VMEM", "VALU cyc"))
    for name, ops in blocks.items():
        s = summarise(ops)
        print("%-10s %5d %5d %5d %4d %5d %9.1f" % (name, s["valu"], s["salu"], s["nops"], s["lds"], s["vmem"], s["cycles"]))
    tot = summarise([o for ops in blocks.values() for o in ops])
    print("%-10s %5d %5d %5d %4d %5d %9.1f   (static total)" % ("all", tot["valu"], tot["salu"], tot["nops"], tot["lds"], tot["vmem"], tot["cycles"]))
    if args.blocks:
        sel = [b.strip() for b in args.blocks.split(",")]
        ops = [o for b in sel for o in blocks[b]]
        s = summarise(ops)
        print("\npath %s:\n  %d VALU, %d SALU, %d s_nop, %d LDS; VALU issue cost %.1f SIMD cycles = %.2f cycles per VALU instruction" % (
            "+".join(sel), s["valu"], s["salu"], s["nops"], s["lds"], s["cycles"], s["cycles"] / max(s["valu"], 1)))
        h = Counter(base_op(o) for o in ops if o.startswith("v_"))
        print("  " + "  ".join("%s x%d (%.1f)" % (k, n, (lookup(k, cost) or DEFAULT_CLASS)) for k, n in h.most_common()))
    if unknown:
        print("\nnot measured, priced at %.1f cycles: %s" % (DEFAULT_CLASS, dict(unknown)))


if __name__ == "__main__":
    main()
