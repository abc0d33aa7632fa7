"""Build libmaest_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(HERE))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmaest_hip.so")
SOURCES = ["capi.hip", "gemm.hip", "gemm256.hip", "gemm_nt_ow.hip", "gemm_tn_ow.hip", "norm.hip", "attention.hip", "attn_fwd_pw.hip", "embed.hip", "misc.hip", "mel.hip", "mel2.hip"]


# sources whose inline asm owns fixed registers: {file: (first, last owned arch VGPR)}; all accumulator registers are owned too
AUDITED = {"attn_fwd_pw.hip": (96, 245), "gemm_nt_ow.hip": (188, 255, True), "gemm_tn_ow.hip": (176, 255, True)}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "maest_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
               "-Wno-unused-result", "-c", s, "-o", o]
        if os.path.basename(s) in AUDITED:
            cmd.insert(-4, "-save-temps=obj")       # keeps the device assembly next to the object for the audit below
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
        if verbose and out.strip():
            sys.stderr.write(out.decode())
    # kernels that own registers by hand: the compiler must have stayed out of them (maest_amd/pw_audit.py)
    from maest_amd import pw_audit
    for name, rng in AUDITED.items():
        lo, hi, regions = rng[0], rng[1], len(rng) > 2
        asm = os.path.join(HERE, "build", os.path.splitext(name)[0] + "-hip-amdgcn-amd-amdhsa-gfx950.s")
        bad, maxv, meta = pw_audit.audit(asm, lo, hi, regions)
        if bad:
            for n, why, st in bad[:20]:
                sys.stderr.write(f"{name}: line {n}: {why}: {st}\n")
            raise RuntimeError(f"{name}: the code object touches registers the kernel owns by hand (or spills)")
        if verbose:
            print(f"audit {name}: compiler's highest arch VGPR v{maxv}, owned v{lo}..v{hi} and the accumulator half untouched; {meta}")
    cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    subprocess.check_call(cmd)
    if verbose:
        print("built", LIB)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
