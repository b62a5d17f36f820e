"""Diagnostic: weight the VALU instructions of an ISA listing (scripts/dbg/isa.sh -> /tmp/isa/k.s) by the issue cost
measured with scripts/dbg/valu_rate.hip (units of one v_xor_b32 slot) and print the cost per line range.
usage: python scripts/dbg/isa_cost.py [file] [first_line last_line] [bucket]"""
import re, sys, collections
FAST = {"v_xor_b32", "v_add_u32", "v_bitop3_b32", "v_fma_f32", "v_mov_b32", "v_and_b32", "v_or_b32", "v_lshrrev_b32", "v_sub_u32", "v_subrev_u32",
        "v_mul_f32", "v_add_f32", "v_not_b32", "v_fmac_f32", "v_sub_f32"}
SLOW = {"v_rcp_f64", "v_rsq_f64", "v_sqrt_f64", "v_trig_preop_f64"}
F64 = {"v_fma_f64", "v_mul_f64", "v_fmac_f64", "v_div_fmas_f64", "v_mul_lo_u32"}
def cost(op):
    base = re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)
    if op.endswith("_dpp"): return 1.65
    if base in FAST: return 1.0
    if base in SLOW: return 6.3
    if base in F64: return 1.85
    return 1.65
def main():
    f = sys.argv[1] if len(sys.argv) > 1 else "/tmp/isa/k.s"
    lines = open(f).read().split("\n")
    a = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    b = min(int(sys.argv[3]) if len(sys.argv) > 3 else len(lines), len(lines))
    bucket = int(sys.argv[4]) if len(sys.argv) > 4 else 50
    tot = 0.0; per = collections.OrderedDict(); byop = collections.Counter(); n = 0
    for i in range(a - 1, b):
        m = re.match(r"\s+(v_\w+)", lines[i])
        if not m: continue
        c = cost(m.group(1)); tot += c; n += 1
        k = (i // bucket) * bucket
        per[k] = per.get(k, 0.0) + c
        byop[re.sub(r"_(e32|e64)$", "", m.group(1))] += c
    print(f"lines {a}-{b}: {n} VALU, cost {tot:.0f} slots")
    for k, v in per.items(): print(f"  {k:5d}-{k+bucket-1:5d}: {v:6.1f}")
    print("by opcode:", ", ".join(f"{o} {v:.0f}" for o, v in byop.most_common(25)))
main()
