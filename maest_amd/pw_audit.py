"""Audit of attn_fwd_pw.hip's code object (the registers that file owns by hand): no scratch, no compiler-generated v_accvgpr_*
or AGPR operand, no arch VGPR of the owned range outside the inline-asm blocks.  Used by maest_amd/build.py; exits non-zero
with the offending lines otherwise.  usage: pw_audit.py <device .s file> <lo> <hi>"""
import re, sys

def audit(path, lo, hi, regions=False):
    """regions = True (gemm_nt_ow.hip): every main-loop asm statement names the owned arch VGPRs as clobbers, so hipcc keeps nothing in them
    across the main loop but may use them elsewhere (the epilogue).  Checked per basic block: a block that holds an MFMA must not
    touch them outside the asm statements (nor scratch), and no SGPR lane spill may sit in one anywhere; a spill of a tile-loop
    invariant in front of the main loop is tolerated."""
    lines = open(path).read().split("\n")
    # basic blocks: a label line opens one
    block_of, has_mfma, blk = [], {}, 0
    for ln in lines:
        if re.match(r"^(\.LBB\w+|_Z\w+):", ln):
            blk += 1
        block_of.append(blk)
        if "v_mfma" in ln:
            has_mfma[blk] = True
    inasm = False
    bad = []
    maxv = -1
    meta = {}
    for n, ln in enumerate(lines, 1):
        if "ASMSTART" in ln:
            inasm = True
            continue
        if "ASMEND" in ln:
            inasm = False
            continue
        m = re.match(r"\s+\.(vgpr_spill_count|sgpr_spill_count|private_segment_fixed_size|vgpr_count|sgpr_count):\s+(\d+)", ln)
        if m:
            meta[m.group(1)] = max(meta.get(m.group(1), 0), int(m.group(2)))
        st = ln.strip()
        if inasm or not st or st[0] in ";." or st.endswith(":"):
            continue
        code = ln.split(";")[0]
        strict = not regions or has_mfma.get(block_of[n - 1], False)
        if "scratch_" in code and strict:
            bad.append((n, "scratch access", st))
        if "v_accvgpr" in code or re.search(r"\ba\d+\b|\ba\[\d", code):
            bad.append((n, "accumulator register outside the asm blocks", st))
        for m in re.finditer(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]", code):
            a = int(m.group(1)) if m.group(1) else int(m.group(2))
            b = int(m.group(1)) if m.group(1) else int(m.group(3))
            owned = b >= lo and a <= hi
            if strict or not owned:
                maxv = max(maxv, b)
            if owned and (strict or "lane_b32" in code):
                bad.append((n, "owned arch VGPR outside the asm blocks", st))
    if not regions and (meta.get("vgpr_spill_count", 0) or meta.get("private_segment_fixed_size", 0)):
        bad.append((0, "spills", str(meta)))
    return bad, maxv, meta


if __name__ == "__main__":
    bad, maxv, meta = audit(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), regions=len(sys.argv) > 4)
    print(f"pw_audit: compiler's highest arch VGPR v{maxv}; {meta}")
    for n, why, st in bad[:20]:
        print(f"  line {n}: {why}: {st}")
    sys.exit(1 if bad else 0)
