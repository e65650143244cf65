"""Builds tests/native/_build/libosfm_ba_emu.so: the product's bundle-adjustment sources (opensfm_amd/csrc/ba.hip with ba_generic.inc / ba_generic_host.inc) compiled
for the HOST against the HIP emulation of tests/native/hipemu -- every kernel and the whole LM driver then run on the CPU.  TEST
INFRASTRUCTURE: nothing under opensfm_amd/ loads this library.

The sources are used as they are, apart from three mechanical substitutions a C++ preprocessor cannot make:
  * `extern __shared__ [attr] T name[];`  ->  `T *name = (T *)hipemu::dyn_lds();`   (dynamic LDS)
  * the inline-assembly LDS barrier `asm volatile("s_waitcnt lgkmcnt(0)\\n\\ts_barrier" ...)`  ->  `__syncthreads()`
  * `#include "x.h"` of the csrc headers keeps working through -I opensfm_amd/csrc
"""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "opensfm_amd", "csrc")
OUT = os.path.join(HERE, "_build")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
SOURCES = ["ba.hip"]


def transform(text: str) -> str:
    text = re.sub(r"extern\s+__shared__\s+(?:__attribute__\(\(aligned\(\d+\)\)\)\s+)?(\w+)\s+(\w+)\[\];",
                  r"\1 *\2 = (\1 *)hipemu::dyn_lds();", text)
    text = re.sub(r'asm volatile\("s_waitcnt lgkmcnt\(0\)\\n\\ts_barrier" ::: "memory"\);', "__syncthreads();", text)
    return text


def build(force: bool = False, sanitize: bool = False) -> str:
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, "libosfm_ba_emu_asan.so" if sanitize else "libosfm_ba_emu.so")
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(HERE, "emu_ctx.cpp"), os.path.abspath(__file__)]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".h", ".inc"))]
    deps += [os.path.join(dp, f) for dp, _, fs in os.walk(os.path.join(HERE, "hipemu")) for f in fs]
    deps.append(os.path.join(ROOT, "include", "osfm_mi355.h"))
    if not force and os.path.exists(so) and all(os.path.getmtime(so) >= os.path.getmtime(d) for d in deps):
        return so
    flags = ["-std=c++17", "-fPIC", "-ffp-contract=off", "-DOSFM_HIPEMU", "-Wno-unknown-attributes", "-Wno-unused-value", "-I", os.path.join(HERE, "hipemu"), "-I", CSRC,
             "-I", os.path.join(ROOT, "include")]
    flags += ["-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"] if sanitize else ["-O2"]
    objs = []
    procs = []
    for inc in os.listdir(CSRC):  # the .inc files the sources include get the same substitutions (found first: next to the generated sources)
        if inc.endswith(".inc"):
            with open(os.path.join(CSRC, inc)) as f:
                text = transform(f.read())
            with open(os.path.join(OUT, inc), "w") as f:
                f.write('#line 1 "%s"\n' % os.path.join(CSRC, inc))
                f.write(text)
    for s in SOURCES:
        gen = os.path.join(OUT, s.replace(".hip", "_emu.cpp"))
        with open(os.path.join(CSRC, s)) as f:
            src = transform(f.read())
        with open(gen, "w") as f:
            f.write('#line 1 "%s"\n' % os.path.join(CSRC, s))
            f.write(src)
        o = gen[:-4] + ("_asan.o" if sanitize else ".o")
        objs.append(o)
        procs.append(subprocess.Popen([CLANG] + flags + ["-c", gen, "-o", o]))
    o = os.path.join(OUT, "emu_ctx_asan.o" if sanitize else "emu_ctx.o")
    objs.append(o)
    procs.append(subprocess.Popen([CLANG] + flags + ["-c", os.path.join(HERE, "emu_ctx.cpp"), "-o", o]))
    for p in procs:
        if p.wait() != 0:
            raise RuntimeError("hipemu build failed")
    subprocess.check_call([CLANG, "-shared", "-fPIC"] + (["-fsanitize=address,undefined"] if sanitize else []) + objs + ["-o", so, "-lpthread"])
    return so


if __name__ == "__main__":
    print(build(force=True, sanitize="--asan" in sys.argv))
