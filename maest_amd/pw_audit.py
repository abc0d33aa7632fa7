"""Audit of attn_fwd_pw.hip's code object (the registers that file owns by hand): no scratch, no compiler-generated v_accvgpr_*
or AGPR operand, no arch VGPR of the owned range outside the inline-asm blocks.  Used by maest_amd/build.py; exits non-zero
with the offending lines otherwise.  usage: pw_audit.py <device .s file> <lo> <hi>"""
import re, sys

def audit(path, lo, hi):
    inasm = False
    bad = []
    maxv = -1
    meta = {}
    for n, ln in enumerate(open(path), 1):
        if "ASMSTART" in ln:
            inasm = True
            continue
        if "ASMEND" in ln:
            inasm = False
            continue
        m = re.match(r"\s+\.(vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|vgpr_count|sgpr_count):\s+(\d+)", ln)
        if m:
            meta[m.group(1)] = int(m.group(2))
        st = ln.strip()
        if inasm or not st or st[0] in ";." or st.endswith(":"):
            continue
        code = ln.split(";")[0]
        if "scratch_" in code:
            bad.append((n, "scratch access", st))
        if "v_accvgpr" in code or re.search(r"\ba\d+\b|\ba\[\d", code):
            bad.append((n, "accumulator register outside the asm blocks", st))
        for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", code):
            a = int(m.group(1)) if m.group(1) else int(m.group(2))
            b = int(m.group(1)) if m.group(1) else int(m.group(3))
            maxv = max(maxv, b)
            if b >= lo and a <= hi:
                bad.append((n, "owned arch VGPR outside the asm blocks", st))
    if meta.get("vgpr_spill_count", 0) or meta.get("private_segment_fixed_size", 0):
        bad.append((0, "spills", str(meta)))
    return bad, maxv, meta

if __name__ == "__main__":
    bad, maxv, meta = audit(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
    print(f"pw_audit: compiler's highest arch VGPR v{maxv}; {meta}")
    for n, why, st in bad[:20]:
        print(f"  line {n}: {why}: {st}")
    sys.exit(1 if bad else 0)
