#!/usr/bin/env python
"""Register / scratch / LDS table of every kernel in vlgp_amd/csrc/*.hip (no GPU needed):

    python tools/resource_usage.py [--all] [file.hip ...] > profiles/r6/resource_usage.txt

compiles each source with `hipcc -Rpass-analysis=kernel-resource-usage` for gfx950 (the flags of the Makefile) and prints
one line per kernel.  Without --all: only the kernels that use scratch, plus every esplit_lane / hstep_round instantiation.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "vlgp_amd", "csrc")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Rpass-analysis=kernel-resource-usage"]
FIELDS = (("VGPR", r"VGPRs: (\d+)"), ("AGPR", r"AGPRs: (\d+)"), ("SGPR", r"SGPRs: (\d+)"),
          ("scratch B/lane", r"ScratchSize \[bytes/lane\]: (\d+)"), ("waves/SIMD", r"Occupancy \[waves/SIMD\]: (\d+)"),
          ("LDS B", r"LDS Size \[bytes/block\]: (\d+)"))


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    return [re.sub(r"\(anonymous namespace\)::", "", ln).split("(")[0].replace("void ", "") for ln in out.stdout.splitlines()]


def usage(path):
    with tempfile.TemporaryDirectory() as tmp:
        done = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-c", path, "-o", os.path.join(tmp, "x.o")],
                              capture_output=True, text=True, cwd=CSRC)
    blocks = re.split(r"remark: [^\n]*Function Name: ", done.stderr)[1:]
    names = demangle([b.split("\n")[0].strip() for b in blocks])
    rows = []
    for name, b in zip(names, blocks):
        vals = []
        for _, pat in FIELDS:
            m = re.search(pat, b)
            vals.append(int(m.group(1)) if m else -1)
        rows.append((name, vals))
    return rows


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    everything = "--all" in sys.argv
    files = [os.path.abspath(a) for a in args] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    print("%-72s %s" % ("kernel", "  ".join("%s" % f for f, _ in FIELDS)))
    for path in files:
        for name, vals in usage(path):
            scratch = vals[3]
            if everything or scratch > 0 or re.search(r"esplit_lane|esplit_mix|esplit_latent|latent_map|hstep_round|mstep_", name):
                print("%-72s %s" % (name[:72], "  ".join("%*d" % (len(f), v) for (f, _), v in zip(FIELDS, vals))))


if __name__ == "__main__":
    main()
