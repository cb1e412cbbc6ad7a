#!/usr/bin/env python3
"""Build experimental / diagnostic code objects of the persistent forward for tools/p4_ab.py (build container, no GPU):

    python3 tools/p4_variants.py base= nolean=P4_LEAN=0 stamp3=P4_STAMP=3 "x=P4_A=1 P4_B=2"

writes photonic_flash_attention_amd/csrc/build/variants/<name>.hsaco (git-ignored; travels to the GPU box with gpurun).  The
generator honours P4_* knobs only with P4_DEV=1, which this script sets; the product `make` never does."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "photonic_flash_attention_amd", "csrc")
OUT = os.path.join(CSRC, "build", "variants")
LLVM = os.environ.get("LLVM", "/opt/rocm/lib/llvm/bin")
os.makedirs(OUT, exist_ok=True)
procs = []
for spec in sys.argv[1:]:
    name, _, envs = spec.partition("=")
    env = {k: v for k, v in os.environ.items() if not k.startswith("P4_")}
    env["P4_DEV"] = "1"
    for kv in envs.split():
        k, _, v = kv.partition("=")
        env[k] = v
    s = os.path.join(OUT, name + ".s")
    with open(s, "w") as f:
        subprocess.run([sys.executable, os.path.join(CSRC, "gen_fa3_fwd_p4.py")], env=env, stdout=f, check=True, cwd=CSRC)
    procs.append((name, subprocess.Popen(
        f"{LLVM}/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c {s} -o {OUT}/{name}.o && "
        f"{LLVM}/ld.lld -shared {OUT}/{name}.o -o {OUT}/{name}.hsaco && rm -f {OUT}/{name}.o", shell=True)))
for name, p in procs:
    rc = p.wait()
    print(f"{name}: {'ok' if rc == 0 else 'FAILED'} -> {OUT}/{name}.hsaco")
    if rc:
        sys.exit(1)
