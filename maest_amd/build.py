"""Build libmaest_hip.so (gfx950) in-tree with hipcc.  Used by __graft_entry__.build()."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(HERE))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmaest_hip.so")
# the same sources with IEEE half as the 16-bit operand type (csrc/common.h: MAEST_16BIT_F16): precision="fp16" evaluation forwards
LIB_F16 = os.path.join(HERE, "libmaest_hip_f16.so")
SOURCES = ["capi.hip", "gemm.hip", "gemm256.hip", "gemm_nt_ow.hip", "gemm_tn_ow.hip", "norm.hip", "attention.hip", "attn_fwd_pw.hip", "embed.hip", "misc.hip", "mel.hip", "mel2.hip"]


# sources whose inline asm owns fixed registers: {file: (first, last owned arch VGPR)}; all accumulator registers are owned too
AUDITED = {"attn_fwd_pw.hip": (96, 245), "gemm_nt_ow.hip": (124, 255, True), "gemm_tn_ow.hip": (176, 255, True)}


def _hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def needs_build():
    if not os.path.exists(LIB) or not os.path.exists(LIB_F16):
        return True
    t = min(os.path.getmtime(LIB), os.path.getmtime(LIB_F16))
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "maest_hip.h"))
    return any(os.path.getmtime(d) > t for d in deps)


VALIDATED_HIPCC = "7.2.26015"    # HIP version of the hipcc the audit was validated with (ROCm 7.2.0, AMD clang 22.0.0git roc-7.2.0)


def hipcc_version():
    try:
        out = subprocess.run([_hipcc(), "--version"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=60).stdout.decode()
    except Exception as e:          # noqa: BLE001
        return f"unknown ({e})"
    for ln in out.splitlines():
        if ln.startswith("HIP version"):
            return ln.split(":", 1)[1].strip()
    return out.splitlines()[0].strip() if out.strip() else "unknown"


def _compile_cmd(src, obj, owned_disabled=False, f16=False):
    cmd = [_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wno-unused-result"]
    if f16:
        cmd.append("-DMAEST_16BIT_F16=1")
    if os.path.basename(src) in AUDITED:
        if owned_disabled:
            cmd.append("-DMAEST_OWNED_DISABLED=1")
        else:
            cmd.append("-save-temps=obj")      # keeps the device assembly next to the object for the audit below
    return cmd + ["-c", src, "-o", obj]


def _device_asm(name, bdir="build"):
    """The gfx950 assembly -save-temps=obj left for `name` (the temp file's name is the compiler's business: glob for it)."""
    import glob
    stem = os.path.splitext(name)[0]
    hits = [f for f in glob.glob(os.path.join(HERE, bdir, stem + "*.s")) if "gfx950" in os.path.basename(f)]
    return max(hits, key=os.path.getmtime) if hits else None


def audit_or_leave_out(name, obj, compile_failed=False, verbose=True, bdir="build", f16=False):
    """Audit the code object of an owned-register source (pw_audit.py); on failure recompile it to `obj` with MAEST_OWNED_DISABLED.
    Returns None when the kernel is in, else the reason it was left out."""
    from maest_amd import pw_audit
    rng = AUDITED[name]
    lo, hi, regions = rng[0], rng[1], len(rng) > 2
    why = None
    if compile_failed:
        why = "hipcc rejected the source"
    else:
        asm = _device_asm(name, bdir)
        if asm is None:
            why = f"no gfx950 assembly (build/{os.path.splitext(name)[0]}*gfx950*.s) left by -save-temps=obj: cannot audit"
        else:
            bad, maxv, meta = pw_audit.audit(asm, lo, hi, regions)
            if bad:
                for n, w, st in bad[:20]:
                    sys.stderr.write(f"{name}: line {n}: {w}: {st}\n")
                why = "the code object touches registers the kernel owns by hand (or spills)"
            elif verbose:
                print(f"audit {name}{' (f16)' if f16 else ''}: compiler's highest arch VGPR v{maxv}, owned v{lo}..v{hi} and the accumulator half untouched; {meta}")
    if why is None:
        return None
    msg = f"{name}: {why} [hipcc {hipcc_version()}; the audit was validated with {VALIDATED_HIPCC}]"
    if os.environ.get("MAEST_STRICT_AUDIT") == "1":
        raise RuntimeError(msg)
    sys.stderr.write("WARNING: " + msg + " -- leaving this kernel out (MAEST_OWNED_DISABLED); the kernel it replaced serves\n")
    cmd = _compile_cmd(os.path.join(CSRC, name), obj, owned_disabled=True, f16=f16)
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        sys.stderr.write(r.stdout.decode())
        raise RuntimeError("hipcc failed: " + " ".join(cmd))
    return why


def build(force=False, verbose=True, leave_out=(), lib=None):
    """Compile every kernel source for gfx950 and link libmaest_hip.so.

    leave_out / lib (tests, `python maest_amd/build.py --leave-out all --lib <path>`): build the fallback form on purpose -- the named
    owned-register sources ("all": every one) compiled with MAEST_OWNED_DISABLED -- into another file, so that the whole GPU suite can be
    run against it (env MAEST_HIP_LIB selects the library maest_amd loads).

    Three sources own fixed registers by hand (AUDITED); their code objects are audited (pw_audit.py).  If an audit cannot run (no
    device assembly found) or fails -- a hipcc that allocates differently from the validated one -- the source is recompiled with
    -DMAEST_OWNED_DISABLED: that kernel is left out, the library dispatches to the kernel it replaced (the eight-wave GEMMs
    / the four-wave attention forward: same results, ~10 % slower), and maest_kernel_forms() reports it.  MAEST_STRICT_AUDIT=1
    turns the fallback into an error (development)."""
    leave_out = set(AUDITED) if "all" in leave_out else set(leave_out)
    if not force and not leave_out and lib is None and not needs_build():
        return LIB
    # both flavours of the 16-bit operand type, compiled side by side (a test build into `lib` is the bf16 flavour only)
    flavours = [(lib or LIB, "build", False)] + ([] if lib is not None else [(LIB_F16, "build_f16", True)])
    import glob
    import json
    jobs = []
    for out_lib, bdir, f16 in flavours:
        srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
        objs = {}
        procs = []
        os.makedirs(os.path.join(HERE, bdir), exist_ok=True)
        for s in srcs:
            if os.path.basename(s) in AUDITED:
                # an assembly file left by an EARLIER build must never be what the audit reads (a hipcc that names its -save-temps output
                # differently, or writes none, would otherwise pass on the old compiler's code): remove it before compiling
                for old in glob.glob(os.path.join(HERE, bdir, os.path.splitext(os.path.basename(s))[0] + "*.s")):
                    os.remove(old)
            o = os.path.join(HERE, bdir, os.path.basename(s) + (".off.o" if os.path.basename(s) in leave_out else ".o"))
            objs[os.path.basename(s)] = o
            cmd = _compile_cmd(s, o, owned_disabled=os.path.basename(s) in leave_out, f16=f16)
            procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        jobs.append((out_lib, bdir, f16, objs, procs))
    for out_lib, bdir, f16, objs, procs in jobs:
        failed = []
        for cmd, p in procs:
            out, _ = p.communicate()
            if p.returncode != 0:
                sys.stderr.write(out.decode())
                if os.path.basename(cmd[-3]) in AUDITED:
                    failed.append(os.path.basename(cmd[-3]))    # e.g. an assembler that rejects the asm: same fallback as a failed audit
                    continue
                raise RuntimeError("hipcc failed: " + " ".join(cmd))
            if verbose and out.strip():
                sys.stderr.write(out.decode())
        left_out = {}
        for name in AUDITED:
            if name in leave_out:
                left_out[name] = "left out on request"
            elif name in objs:
                why = audit_or_leave_out(name, objs[name], compile_failed=name in failed, verbose=verbose, bdir=bdir, f16=f16)
                if why:
                    left_out[name] = why
        cmd = [_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out_lib] + list(objs.values())
        subprocess.check_call(cmd)
        with open(os.path.join(HERE, bdir, "build_info.json" if lib is None else os.path.basename(out_lib) + ".build_info.json"), "w") as f:
            json.dump({"hipcc": hipcc_version(), "validated_with": VALIDATED_HIPCC, "left_out": left_out, "flavour": "f16" if f16 else "bf16"}, f, indent=1)
        if verbose:
            print("built", out_lib, "(all owned-register kernels in)" if not left_out else f"(left out: {sorted(left_out)})")
    return lib or LIB


if __name__ == "__main__":
    a = sys.argv[1:]
    lo = a[a.index("--leave-out") + 1].split(",") if "--leave-out" in a else ()
    build(force="--force" in a, leave_out=lo, lib=a[a.index("--lib") + 1] if "--lib" in a else None)
