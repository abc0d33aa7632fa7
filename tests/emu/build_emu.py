"""Build tests/emu/libmaest_emu.so: the SAME kernel sources (maest_amd/csrc/*.hip) compiled for the
host with the SIMT lockstep emulator shadowing <hip/hip_runtime.h>.  TEST INFRASTRUCTURE ONLY."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(REPO, "maest_amd", "csrc")
LIB = os.path.join(HERE, "libmaest_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def available():
    return os.path.exists(CLANG)


def build(force=False):
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    deps = srcs + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    deps += [os.path.join(HERE, "include", "hip", "hip_runtime.h"), os.path.join(HERE, "emu_runtime.cpp"),
             os.path.join(REPO, "include", "maest_hip.h")]
    if not force and os.path.exists(LIB) and all(os.path.getmtime(d) <= os.path.getmtime(LIB) for d in deps):
        return LIB
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    objs, procs = [], []
    for s in srcs + [os.path.join(HERE, "emu_runtime.cpp")]:
        o = os.path.join(HERE, "build", os.path.basename(s) + ".o")
        objs.append(o)
        cmd = [CLANG, "-x", "c++", "-std=c++20", "-O2", "-fPIC", "-pthread", "-ffp-contract=off",
               "-I", os.path.join(HERE, "include"), "-Wno-unused-value", "-c", s, "-o", o]
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(out.decode())
            raise RuntimeError("emu build failed: " + " ".join(cmd))
    subprocess.check_call([CLANG, "-shared", "-fPIC", "-pthread", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
